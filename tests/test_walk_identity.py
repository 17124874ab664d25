"""The identity the agglomeration on the GPU rests on (swarm_amd/csrc/cluster_gpu.hip, cluster_dn.cpp: cluster_on_device),
checked on the CPU by brute force: the reference's greedy walk over the graph of accepted pairs (src/algo.cc:384-602 for
d >= 2, src/algod1.cc:1175-1257 for d = 1: seeds by lowest unswarmed id, a breadth-first queue per swarm kept in
(generation, id) order — find_correct_position_in_list, src/algo.cc:205-219 — a sub-seed's hits taken in id order) is a
pure function of the directed graph:

    swarm(v)      = the smallest id that reaches v
    generation(v) = its distance from that seed (inside the swarm)
    parent(v)     = the smallest id of the previous generation that points at v
    members       in (generation, id) order;  radius(v) = radius(parent) + diff(parent, v)
    -i lines      a swarm's members grouped by their parent's place in that order, a parent's hits by id

Random directed graphs with the shape the abundance rule gives (links from lower to higher ids, both directions where
abundances tie), differences 1..3 per link."""
import numpy as np


def _graph(rng, n, density, ties):
    """rows[u] = sorted list of (v, diff): u may take v.  Amplicons are in abundance order, so u < v unless they tie."""
    group = np.cumsum(rng.random(n) > ties)                 # equal numbers = tied abundances
    rows = [dict() for _ in range(n)]
    m = int(density * n)
    for _ in range(m):
        u, v = (int(x) for x in rng.integers(0, n, 2))
        if u == v:
            continue
        d = int(rng.integers(1, 4))
        lo, hi = min(u, v), max(u, v)
        rows[lo][hi] = d
        if group[lo] == group[hi]:
            rows[hi][lo] = d
    return [sorted(r.items()) for r in rows]


def _greedy_walk(rows):
    """cluster_over_graph (swarm_amd/csrc/host/cluster_dn.cpp), the loop of the reference restated"""
    n = len(rows)
    swarmed = [False] * n
    swarms, links = [], []
    for seed in range(n):
        if swarmed[seed]:
            continue
        swarmed[seed] = True
        queue = [(seed, 0, 0)]                               # (id, generation, radius)
        nxt = 0
        while nxt < len(queue):
            sid, sgen, srad = queue[nxt]
            nxt += 1
            for v, d in rows[sid]:
                if swarmed[v]:
                    continue
                swarmed[v] = True
                pos = len(queue)
                while pos > nxt and queue[pos - 1][0] > v and queue[pos - 1][1] > sgen:
                    pos -= 1
                queue.insert(pos, (v, sgen + 1, srad + d))
                links.append((sid, v, d, len(swarms) + 1, sgen + 1))
        swarms.append(queue)
    return swarms, links


def _functional_form(rows):
    """what k_label_step / k_level_step / the sort compute, and what cluster_on_device builds from it"""
    n = len(rows)
    label = list(range(n))
    changed = True
    while changed:                                           # label[v] = min over the links u -> v
        changed = False
        for u in range(n):
            for v, _ in rows[u]:
                if label[u] < label[v]:
                    label[v] = label[u]
                    changed = True
    gen = [0 if label[v] == v else None for v in range(n)]
    parent = [None] * n
    level = 1
    while True:                                              # level-synchronous claims inside the swarm
        grew = False
        claims = {}
        for u in range(n):
            if gen[u] != level - 1:
                continue
            for v, _ in rows[u]:
                if label[v] == label[u] and gen[v] is None:
                    claims[v] = min(claims.get(v, u), u)
        for v, u in claims.items():
            gen[v], parent[v] = level, u
            grew = True
        if not grew:
            break
        level += 1
    seeds = [v for v in range(n) if label[v] == v]
    swarm_no = {s: i for i, s in enumerate(seeds)}
    order = sorted(range(n), key=lambda v: (swarm_no[label[v]], gen[v], v))
    diff = [dict(r) for r in rows]
    radius = [0] * n
    swarms, links = [[] for _ in seeds], []
    place = {}
    for v in order:
        s = swarm_no[label[v]]
        if gen[v]:
            radius[v] = radius[parent[v]] + diff[parent[v]][v]
        place[v] = len(swarms[s])
        swarms[s].append((v, gen[v], radius[v]))
    for s, members in enumerate(swarms):                     # stable sort of the non-seed members by their parent's place
        kids = sorted(members[1:], key=lambda m: place[parent[m[0]]])
        links += [(parent[v], v, diff[parent[v]][v], s + 1, g) for v, g, _ in kids]
    return swarms, links


def test_greedy_walk_is_a_function_of_the_directed_graph():
    rng = np.random.default_rng(23)
    checked = 0
    for t in range(300):
        n = int(rng.integers(1, 120))
        rows = _graph(rng, n, float(rng.choice([0.3, 0.8, 1.5, 3.0])), float(rng.choice([0.0, 0.3, 0.7])))
        want_swarms, want_links = _greedy_walk(rows)
        got_swarms, got_links = _functional_form(rows)
        assert got_swarms == want_swarms, (t, rows)
        assert got_links == want_links, (t, rows)
        checked += sum(len(s) > 2 for s in want_swarms)
    assert checked > 300                                     # swarms with several generations were among them
