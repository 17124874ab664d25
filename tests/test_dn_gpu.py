"""d >= 2 end to end on the GPU: FASTA -> HostDb -> the search (B3 + B4) -> greedy agglomeration -> writers,
byte-compared with the reference's own output files.  Two routes for the search, both exercised:
"graph" = every pair within d differences at once (dn_graph.hip; taken when every sequence has room for
d + 1 windows; the greedy walk then runs on the GPU as well — or, "graph_host_walk", on the downloaded graph),
"scan" = one fused scan step per swarm generation (scan.hip) driving the host's loop."""
import filecmp

import numpy as np
import pytest

import support as S
from swarm_amd import DnClusters, HostDb

pytestmark = pytest.mark.gpu
G = S.GOLDEN


@pytest.fixture(params=["auto", "scan", "graph_host_walk"])
def route(request, monkeypatch):
    monkeypatch.delenv("SWARM_AMD_DN_WALK", raising=False)
    if request.param == "scan":
        monkeypatch.setenv("SWARM_AMD_DN", "scan")
    else:
        monkeypatch.delenv("SWARM_AMD_DN", raising=False)
        if request.param == "graph_host_walk":
            monkeypatch.setenv("SWARM_AMD_DN_WALK", "host")
    return request.param


@pytest.mark.parametrize("name", ["d2_small", "d3_400", "d5_ties", "d8_16bit"])
def test_dn_outputs_byte_identical(gpu_ctx, tmp_path, name, route):
    args = (G / f"{name}.args").read_text().split()
    d = int(args[args.index("-d") + 1])
    hdb = HostDb(G / f"{name}.fasta", check_duplicate_sequences=True)
    gpu_ctx.upload_hostdb(hdb)
    cl = DnClusters(gpu_ctx, hdb, d)
    cl.write_swarms(tmp_path / "o")
    assert filecmp.cmp(tmp_path / "o", G / f"{name}.o", shallow=False)
    for suffix, writer in (("s", cl.write_stats), ("i", cl.write_structure), ("w", cl.write_seeds), ("u", cl.write_uclust)):
        if (G / f"{name}.{suffix}").exists():
            writer(tmp_path / suffix)
            assert filecmp.cmp(tmp_path / suffix, G / f"{name}.{suffix}", shallow=False), suffix
    log = (G / f"{name}.log").read_text()
    s = cl.summary()
    assert f"Number of swarms:  {s['swarms']}\n" in log
    assert f"Largest swarm:     {s['largest']}\n" in log
    assert f"Max generations:   {s['maxgen']}\n" in log
    t = cl.scan_totals()
    assert t["qgram_comparisons"] > 0 and t["aligned_pairs"] > 0
    if route == "scan":
        assert t["route"] == "scan"
    elif name in ("d2_small", "d3_400"):              # 60 nt, d = 2 / 120 nt, d = 3: room for d + 1 windows of 16
        assert t["route"] == "graph"


@pytest.mark.skipif(not S.have_reference(), reason="compiled reference not available on this box")
@pytest.mark.parametrize("n,length,d,edits,extra", [(3000, 150, 2, 2, []), (2000, 400, 3, 3, []), (1500, 100, 3, 3, ["-n"]),
                                                     # bigger sets: long generation chains (radius outgrows the candidate
                                                     # list's bound), many swarms, the wavefront kernel at d = 2 and 3, the
                                                     # banded one at d = 4
                                                     (25000, 200, 2, 2, []), (15000, 300, 3, 3, []), (8000, 120, 4, 4, []),
                                                     (6000, 60, 3, 3, ["-n"])])
def test_dn_against_reference_binary(gpu_ctx, tmp_path, n, length, d, edits, extra, route):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, length, 900 + d, edits)
    r = S.run_ref_swarm(["-d", d, "-o", tmp_path / "ro", "-i", tmp_path / "ri", "-s", tmp_path / "rs", "-l", "/dev/null"]
                        + extra + [fa])
    assert r.returncode == 0, r.stderr
    hdb = HostDb(fa, check_duplicate_sequences=True)
    gpu_ctx.upload_hostdb(hdb)
    cl = DnClusters(gpu_ctx, hdb, d, no_cluster_breaking="-n" in extra)
    cl.write_swarms(tmp_path / "o")
    cl.write_structure(tmp_path / "i")
    cl.write_stats(tmp_path / "s")
    for suffix in "ois":
        assert filecmp.cmp(tmp_path / suffix, tmp_path / ("r" + suffix), shallow=False), suffix


def _tied_low_complexity_set(path, seed, families, length, d):
    """Families over a two-letter alphabet with long runs (windows that repeat at several shifts, many equally
    good alignments), members 0..d+1 edits from their centroid, abundances drawn from {1, 1, 1, 2, 3}: most pairs
    tie on abundance, so both directions of a pair are alignments the walk may need."""
    rng = np.random.default_rng(seed)
    seen, recs = set(), []
    for f in range(families):
        cent = []
        while len(cent) < length:
            cent += [str(rng.choice(list("AC")))] * int(rng.integers(1, 7))
        cent = "".join(cent[:length + int(rng.integers(-2, 3))])
        for m in range(14):
            s = cent
            for _ in range(int(rng.integers(0, d + 2)) if m else 0):
                p = int(rng.integers(0, len(s)))
                k = int(rng.integers(0, 3))
                b = str(rng.choice(list("ACGT")))
                s = s[:p] + b + s[p + 1:] if k == 0 else (s[:p] + s[p + 1:] if k == 1 else s[:p] + b + s[p:])
            if s not in seen:
                seen.add(s)
                recs.append((f"f{f}m{m}_{int(rng.choice([1, 1, 1, 2, 3]))}", s))
    path.write_text("".join(f">{h}\n{s}\n" for h, s in recs))


@pytest.mark.skipif(not S.have_reference(), reason="compiled reference not available on this box")
@pytest.mark.parametrize("length,d,extra", [(140, 2, []), (140, 3, []), (200, 3, ["-n"]), (70, 2, []), (270, 4, [])])
def test_dn_graph_route_on_tied_low_complexity_sets(gpu_ctx, tmp_path, monkeypatch, length, d, extra):
    monkeypatch.setenv("SWARM_AMD_DN", "graph")       # must be taken (and is an error if it cannot be)
    fa = tmp_path / "in.fa"
    _tied_low_complexity_set(fa, 100 + d, 400, length, d)
    r = S.run_ref_swarm(["-d", d, "-o", tmp_path / "ro", "-i", tmp_path / "ri", "-s", tmp_path / "rs", "-l", "/dev/null"]
                        + extra + [fa])
    assert r.returncode == 0, r.stderr
    hdb = HostDb(fa, check_duplicate_sequences=True)
    gpu_ctx.upload_hostdb(hdb)
    cl = DnClusters(gpu_ctx, hdb, d, no_cluster_breaking="-n" in extra)
    assert cl.scan_totals()["route"] == "graph"
    cl.write_swarms(tmp_path / "o")
    cl.write_structure(tmp_path / "i")
    cl.write_stats(tmp_path / "s")
    for suffix in "ois":
        assert filecmp.cmp(tmp_path / suffix, tmp_path / ("r" + suffix), shallow=False), suffix
