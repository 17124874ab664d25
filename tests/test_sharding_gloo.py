"""The N > 1 path on CPU: world_size-2 (and 3) gloo processes shard the query range, each
computes the CSR of its slice (here with the oracle standing in for the GPU kernel) and the
slices are all-gathered by swarm_amd.sharding exactly as bench.py does over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import support as S
from swarm_amd import sharding

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
    import support as S
    from swarm_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    db = S.db_from_fasta(sys.argv[2])
    parts = sharding.partition_by_length(db.seqlen, world) if sys.argv[3] == "length" else sharding.partition_even(db.n, world)
    first, count = parts[rank]
    off, nb, _ = S.oracle_d1_network(db, False, first, count)
    nb = nb.copy()
    for i in range(count):
        nb[int(off[i]):int(off[i + 1])].sort()
    t_off = torch.from_numpy(off.astype(np.int64))
    t_nb = torch.from_numpy(np.concatenate([nb, np.zeros(5, dtype=np.uint32)]).view(np.int32))   # capacity > total
    g_off, g_nb = sharding.allgather_csr(t_off, t_nb, len(nb), [c for _, c in parts])
    full_off, full_nb, _ = S.oracle_d1_network(db)
    full_nb = full_nb.copy()
    for i in range(db.n):
        full_nb[int(full_off[i]):int(full_off[i + 1])].sort()
    assert np.array_equal(g_off.numpy().astype(np.uint64), full_off), "offsets differ"
    assert np.array_equal(g_nb.numpy().view(np.uint32), full_nb), "neighbours differ"
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _star_fasta(path, arms=300):
    """one centre with `arms` substitution neighbours: a row of more than 255 links (row counts then travel as 32-bit
    integers, not bytes)"""
    rng = np.random.default_rng(3)
    centre = "".join(rng.choice(list("ACGT"), 120))
    seqs = {centre}
    while len(seqs) < arms + 1:
        p = int(rng.integers(0, len(centre)))
        seqs.add(centre[:p] + "ACGT"[("ACGT".index(centre[p]) + int(rng.integers(1, 4))) % 4] + centre[p + 1:])
    recs = [(centre, 1000)] + [(s, 1 + i % 7) for i, s in enumerate(sorted(seqs - {centre}))]
    path.write_text("".join(f">s{i}_{a}\n{s}\n" for i, (s, a) in enumerate(recs)))


@pytest.mark.parametrize("world,mode", [(2, "even"), (2, "length"), (3, "length"), (2, "star"), (4, "even")])
def test_sharded_network_equals_single(tmp_path, world, mode):
    fa = tmp_path / "in.fa"
    if mode == "star":
        _star_fasta(fa)
        mode = "even"
    else:
        S.gen_fasta(fa, 3001, 90, 77)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script), str(S.ROOT),
                        str(fa), mode], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


def test_partitions_cover_range():
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 1000):
        lens = rng.integers(1, 500, size=n).astype(np.uint32)
        for world in (1, 2, 3, 8):
            for parts in (sharding.partition_even(n, world), sharding.partition_by_length(lens, world)):
                assert len(parts) == world
                assert parts[0][0] == 0 and sum(c for _, c in parts) == n
                for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                    assert f0 + c0 == f1
    lens = np.full(1000, 150, dtype=np.uint32)
    assert all(abs(c - 125) <= 1 for _, c in sharding.partition_by_length(lens, 8))


GRAFT_WORKER = textwrap.dedent('''
    import sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    from swarm_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    NO = 0xFFFFFFFF
    rng = np.random.default_rng(5)                       # same stream on every rank
    n = 5000
    full = np.where(rng.random(n) < 0.3, rng.integers(0, n, size=n), NO).astype(np.uint32)
    owner = rng.integers(0, world, size=n)               # the rank whose heavy slice holds the minimum
    worse = np.minimum(full.astype(np.int64) + rng.integers(1, 50, size=n), NO - 1).astype(np.uint32)
    other = np.where(rng.random(n) < 0.5, worse, NO).astype(np.uint32)   # a larger candidate or none
    mine = np.where(full == NO, NO, np.where(owner == rank, full, other)).astype(np.uint32)
    counters = np.array([1234, 100 + rank, 7 * (rank + 1), 4096, 6, 0, 0, 0], dtype=np.uint64)
    g, c = sharding.combine_grafts(mine, counters)
    assert g.dtype == np.uint32 and np.array_equal(g, full), "graft minimum differs"
    assert int(c[1]) == sum(100 + r for r in range(world)) and int(c[2]) == sum(7 * (r + 1) for r in range(world))
    assert [int(c[0]), int(c[3]), int(c[4])] == [1234, 4096, 6]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


@pytest.mark.parametrize("world", [2, 3])
def test_combine_grafts_min_and_sum(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(GRAFT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script), str(S.ROOT)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


OWNED_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
    import support as S
    from swarm_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    db = S.db_from_fasta(sys.argv[2])
    full_off, full_nb, _ = S.oracle_d1_network(db)
    full_nb = full_nb.copy()
    for i in range(db.n):
        full_nb[int(full_off[i]):int(full_off[i + 1])].sort()
    # what a rank of an ownership-sharded job holds: a flat list of links, every link of the network
    # on exactly one rank (here: a hash of the link stands in for "the rank owning the anchor group")
    rows = np.repeat(np.arange(db.n, dtype=np.uint64), np.diff(full_off).astype(np.int64))
    owner = ((rows * np.uint64(2654435761) + full_nb.astype(np.uint64) * np.uint64(40503)) >> np.uint64(7)) % np.uint64(world)
    keep = owner == rank
    links = ((rows[keep] << np.uint64(32)) | full_nb[keep].astype(np.uint64)).astype(np.int64)
    links = links[np.random.default_rng(rank).permutation(len(links))]        # the kernels emit them unordered
    parts = sharding.partition_by_length(db.seqlen, world) if sys.argv[3] == "length" else sharding.partition_even(db.n, world)
    counts = [c for _, c in parts]
    l_off, l_nb = sharding.exchange_owned_links(torch.from_numpy(links), counts)
    first, count = parts[rank]
    lo, hi = int(full_off[first]), int(full_off[first + count])
    assert np.array_equal(l_off.numpy().astype(np.uint64), full_off[first:first + count + 1] - full_off[first]), "slice offsets differ"
    assert np.array_equal(l_nb.numpy().view(np.uint32), full_nb[lo:hi]), "slice neighbours differ"
    g_off, g_nb = sharding.allgather_csr(l_off, l_nb, int(l_nb.numel()), counts)
    assert np.array_equal(g_off.numpy().astype(np.uint64), full_off), "offsets differ"
    assert np.array_equal(g_nb.numpy().view(np.uint32), full_nb), "neighbours differ"
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


@pytest.mark.parametrize("world,mode", [(2, "even"), (3, "length"), (4, "even"), (8, "length")])   # (8: what the driver's scaling run starts)
def test_owned_links_merge_into_the_network(tmp_path, world, mode):
    """Ownership sharding (swa_d1_set_ownership): every rank holds some of the links (flat list),
    all-to-all by seed range, CSR slice per rank, then the usual all-gather of the slices."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 2503, 90, 78)
    script = tmp_path / "worker.py"
    script.write_text(OWNED_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script), str(S.ROOT),
                        str(fa), mode], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


ROUTE_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    from swarm_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 10007
    owner = lambda ids, index: (ids * (2654435761 if index == 0 else 40503) >> 7) % world        # stands in for the key's owner
    parts = sharding.partition_even(n, world)
    first, count = parts[rank]
    cap = 3 * count // (2 * world) + 1024
    d_ids = torch.zeros(2 * world * cap, dtype=torch.int32)
    d_counts = torch.zeros(2 * world + 1, dtype=torch.int32)
    mine = np.arange(first, first + count, dtype=np.int64)
    for index in range(2):
        own = owner(mine, index)
        for o in range(world):
            sel = mine[own == o]
            k = index * world + o
            d_ids[k * cap: k * cap + len(sel)] = torch.from_numpy(sel.astype(np.int32))
            d_counts[k] = len(sel)
    got = sharding.exchange_routed_ids(d_ids, d_counts, cap)
    everybody = np.arange(n, dtype=np.int64)
    for index in range(2):
        want = everybody[owner(everybody, index) == rank]                      # sources in rank order, ids ascending inside
        assert np.array_equal(got[index].numpy().astype(np.int64), want), (rank, index)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


RECORD_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    from swarm_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 10007
    owner = lambda ids, index: (ids * (2654435761 if index == 0 else 40503) >> 7) % world        # stands in for the key's owner
    key = lambda ids, index: (ids * 0x9E3779B1 + index) & 0xFFFFFFF8                              # ... for the record key
    parts = sharding.partition_even(n, world)
    first, count = parts[rank]
    cap = 3 * count // (2 * world) + 1024
    d_rec = torch.zeros(2 * world * cap, dtype=torch.int64)
    d_counts = torch.zeros(2 * world + 1, dtype=torch.int32)
    mine = np.arange(first, first + count, dtype=np.int64)
    for index in range(2):
        own = owner(mine, index)
        for o in range(world):
            sel = mine[own == o]
            k = index * world + o
            d_rec[k * cap: k * cap + len(sel)] = torch.from_numpy((key(sel, index) << 32) | sel)
            d_counts[k] = len(sel)
    rec_p, rec_s = sharding.exchange_routed_records(d_rec, d_counts, cap)
    everybody = np.arange(n, dtype=np.int64)
    for index, got in ((0, rec_p), (1, rec_s)):
        want = everybody[owner(everybody, index) == rank]                      # sources in rank order, ids ascending inside
        assert np.array_equal(got.numpy(), (key(want, index) << 32) | want), (rank, index)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


@pytest.mark.parametrize("world", [2, 3, 8])
def test_routed_records_reach_their_owners(tmp_path, world):
    """sharding.exchange_routed_records: what swa_d1_route_slice_records leaves on every rank (key records of its slice by
    owning rank and index) arrives, all-to-all, as the arguments of swa_d1_index_build_records."""
    script = tmp_path / "record_worker.py"
    script.write_text(RECORD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script), str(S.ROOT)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


@pytest.mark.parametrize("world", [2, 3, 8])
def test_routed_ids_reach_their_owners(tmp_path, world):
    """sharding.exchange_routed_ids: what swa_d1_route_slice leaves on every rank (ids of its slice by owning rank and
    index) arrives, all-to-all, as the member lists of swa_d1_index_build_routed."""
    script = tmp_path / "route_worker.py"
    script.write_text(ROUTE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script), str(S.ROOT)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


DN_WORKER = textwrap.dedent('''
    import sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    from swarm_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # a d >= 2 graph (same stream on every rank): n amplicons, directed pairs (query, target, diff) within d differences;
    # the rank that owns a pair = the rank owning the first window group the two share (here: a hash of the pair)
    rng = np.random.default_rng(9)
    n, m = 4001, 30011
    q = rng.integers(0, n, size=m).astype(np.uint64)
    t = rng.integers(0, n, size=m).astype(np.uint64)
    keep = q != t
    keys = np.unique((q[keep] << np.uint64(32)) | t[keep])            # every pair once, sorted
    diffs = ((keys * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(61)).astype(np.uint8) % 3 + 1
    owner = ((keys * np.uint64(2654435761)) >> np.uint64(13)) % np.uint64(world)
    if len(sys.argv) > 2 and sys.argv[2] == "empty_rank":
        owner[owner == world - 1] = 0                                  # a rank that owns no group at all
    sel = owner == rank
    mine_k = torch.from_numpy(keys[sel].astype(np.int64))             # sorted: what swa_dn_graph_compute leaves in HBM
    mine_d = torch.from_numpy(diffs[sel])
    got = sharding.gather_dn_graph(mine_k, mine_d, n)
    if rank == 0:
        off, nb, df = got
        want_off = np.searchsorted(keys, np.arange(n + 1, dtype=np.uint64) << np.uint64(32))
        assert np.array_equal(off.numpy(), want_off), "offsets differ"
        assert np.array_equal(nb.numpy().view(np.uint32), (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)), "neighbours differ"
        assert np.array_equal(df.numpy(), diffs), "differences differ"
        rows_sorted = all(np.all(np.diff(nb.numpy().view(np.uint32)[want_off[i]:want_off[i + 1]].astype(np.int64)) > 0) for i in range(0, n, 97))
        assert rows_sorted
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


@pytest.mark.parametrize("world,mode", [(2, "plain"), (3, "plain"), (3, "empty_rank")])
def test_dn_triples_merge_into_the_graph(tmp_path, world, mode):
    """The d >= 2 exchange (second paragraph; swa_multi_dn_graph in C++): every rank holds the sorted triples of
    the window groups it owns, rank 0 gathers and MERGES them (no sort of the whole) into the CSR the greedy walk reads."""
    script = tmp_path / "dn_worker.py"
    script.write_text(DN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script), str(S.ROOT), mode],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


def test_merge_of_sorted_lists_equals_a_sort():
    import torch
    rng = np.random.default_rng(4)
    for world in (1, 2, 5):
        keys = np.unique(rng.integers(0, 1 << 40, size=5000).astype(np.int64))
        owner = rng.integers(0, world, size=len(keys))
        lists = [torch.from_numpy(keys[owner == r]) for r in range(world)]
        pays = [torch.from_numpy((keys[owner == r] % 251).astype(np.uint8)) for r in range(world)]
        k, p = sharding.merge_sorted_lists(lists, pays)
        assert np.array_equal(k.numpy(), keys) and np.array_equal(p.numpy(), (keys % 251).astype(np.uint8))
