"""B2 parity on the GPU: swa_d1_fastidious (HIP) vs the oracle and vs the reference's own
output files (tests/golden), through the C ABI; the host clustering consumes the HIP path's
neighbour lists here."""
import filecmp
import re

import numpy as np
import pytest

import support as S
from swarm_amd import D1Clusters, HostDb

pytestmark = pytest.mark.gpu
G = S.GOLDEN


def _pipeline(ctx, fasta, boundary, bits):
    hdb = HostDb(fasta)
    ctx.upload_hostdb(hdb)
    assert ctx.d1_index_build() is False
    off, nb = ctx.d1_network()
    cl = D1Clusters(hdb, off, nb)
    flags, stats = cl.light_flags(boundary)
    graft, counters = ctx.d1_fastidious(flags, stats[2], bits)
    return hdb, cl, flags, stats, graft, counters


@pytest.mark.parametrize("name,boundary,bits", [("d1_fastidious", 3, 16), ("d1_fastidious_b10_y8", 10, 8)])
def test_fastidious_matches_reference_outputs(gpu_ctx, tmp_path, name, boundary, bits):
    hdb, cl, flags, stats, graft, counters = _pipeline(gpu_ctx, G / f"{name}.fasta", boundary, bits)
    log = (G / f"{name}.log").read_text()
    pick = lambda pat: int(re.search(pat, log).group(1))
    assert int(counters[0]) == pick(r"Generated (\d+) variants")
    assert int(counters[1]) == pick(r"Heavy variants: (\d+)")
    assert int(counters[2]) == pick(r"Got (\d+) graft")
    assert (int(counters[3]), int(counters[4])) == (pick(r"m=(\d+)"), pick(r"k=(\d+)"))
    db = S.db_from_fasta(G / f"{name}.fasta")
    want_graft, want_counters = S.oracle_fastidious(db, flags, bits)
    assert np.array_equal(graft, want_graft)
    assert cl.graft(graft) == pick(r"Made (\d+) grafts")
    cl.write_swarms(tmp_path / "o")
    cl.write_stats(tmp_path / "s")
    cl.write_structure(tmp_path / "i")
    for suffix in "osi":
        assert filecmp.cmp(tmp_path / suffix, G / f"{name}.{suffix}", shallow=False), suffix


# lengths 112 / 114: sequences on both sides of the pair route's minimum length (d1_fast.inc: kFastMinLen), so that
# both routes run and split the pairs between them; 400: the pair route's larger LDS sets
@pytest.mark.parametrize("n,length,seed,light,boundary,bits", [(20000, 150, 41, 0.3, 3, 16), (6000, 400, 42, 0.2, 3, 16),
                                                            (8000, 40, 43, 0.4, 5, 4), (3000, 150, 44, 0.0, 3, 16),
                                                            (15000, 112, 45, 0.3, 3, 16), (15000, 114, 46, 0.35, 3, 16),
                                                            (60000, 150, 47, 0.3, 3, 16)])
def test_fastidious_matches_oracle(gpu_ctx, tmp_path, n, length, seed, light, boundary, bits):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, length, seed, 1, light)
    hdb, cl, flags, stats, graft, counters = _pipeline(gpu_ctx, fa, boundary, bits)
    db = S.db_from_fasta(fa)
    want_graft, want_counters = S.oracle_fastidious(db, flags, bits)
    assert np.array_equal(graft, want_graft)
    assert [int(x) for x in counters[:5]] == [int(x) for x in want_counters[:5]]


def _low_complexity_set(path, seed, families=300, length=140):
    """Families of sequences over a two-letter alphabet with long homopolymer runs, 0-3 random edits away
    from their centroid: many ways to align a pair, many coinciding microvariants — the hard case for
    'within two edits' and for counting |V1(h) & V1(x)|.  Abundance 1 (light) for two thirds of them."""
    rng = np.random.default_rng(seed)
    seen, recs = set(), []
    for f in range(families):
        cent = []
        while len(cent) < length:
            cent += [str(rng.choice(list("AC")))] * int(rng.integers(1, 9))
        cent = "".join(cent[:length + int(rng.integers(-3, 4))])
        for m in range(12):
            s = cent
            for _ in range(int(rng.integers(0, 4)) if m else 0):
                p = int(rng.integers(0, len(s)))
                k = int(rng.integers(0, 3))
                b = str(rng.choice(list("ACG")))
                s = s[:p] + b + s[p + 1:] if k == 0 else (s[:p] + s[p + 1:] if k == 1 else s[:p] + b + s[p:])
            if s not in seen:
                seen.add(s)
                recs.append((f"f{f}m{m}_{100 + f if m == 0 else (1 if m % 3 else 5)}", s))
    path.write_text("".join(f">{h}\n{s}\n" for h, s in recs))


@pytest.mark.parametrize("seed,length", [(7, 140), (8, 113), (9, 200)])
def test_fastidious_low_complexity_families(gpu_ctx, tmp_path, seed, length):
    fa = tmp_path / "low.fa"
    _low_complexity_set(fa, seed, length=length)
    hdb, cl, flags, stats, graft, counters = _pipeline(gpu_ctx, fa, 3, 16)
    db = S.db_from_fasta(fa)
    want_graft, want_counters = S.oracle_fastidious(db, flags, 16)
    assert int(want_counters[2]) > 1000
    assert np.array_equal(graft, want_graft)
    assert [int(x) for x in counters[:5]] == [int(x) for x in want_counters[:5]]


def test_pair_route_equals_bloom_route(tmp_path, monkeypatch):
    """The same seam computed by the reference's own scheme (Bloom filter of light microvariants, SWA_FAST_BLOOM=1)
    and by the pair route: identical graft candidates and counters."""
    from swarm_amd import Context
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 30000, 130, 48, 1, 0.3)
    res = []
    for forced in ("0", "1"):
        monkeypatch.setenv("SWA_FAST_BLOOM", forced)
        ctx = Context(0)
        hdb, cl, flags, stats, graft, counters = _pipeline(ctx, fa, 3, 16)
        res.append((graft, [int(x) for x in counters[:5]]))
        ctx.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert (res[0][0] != 0xFFFFFFFF).sum() > 1000


def test_pair_kernel_on_lines_equals_the_one_on_packed_words(tmp_path, monkeypatch):
    """k_fast_pairs_lines (both sequences in registers, pairs of 64 groups dealt to the lanes: round 3) against round 2's
    k_fast_pairs (SWA_FAST_PAIRS=words), on a set with low-complexity families: identical graft candidates and counters."""
    from swarm_amd import Context
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 40000, 140, 77, 1, 0.3)
    res = []
    for route in ("lines", "words"):
        monkeypatch.setenv("SWA_FAST_PAIRS", route)
        ctx = Context(0)
        hdb, cl, flags, stats, graft, counters = _pipeline(ctx, fa, 3, 16)
        res.append((graft, [int(x) for x in counters[:5]]))
        ctx.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert (res[0][0] != 0xFFFFFFFF).sum() > 1000


@pytest.mark.parametrize("nshards", [2, 3])
def test_fastidious_shards_combine_to_whole(gpu_ctx, tmp_path, nshards):
    """heavy amplicons split over GPUs; minimum of graft_cand, sum of the heavy
    counters == the unsharded pass (each shard run on this one GPU in turn)."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 12000, 120, 51, 1, 0.3)
    hdb, cl, flags, stats, graft, counters = _pipeline(gpu_ctx, fa, 3, 16)
    assert (graft != 0xFFFFFFFF).sum() > 50
    merged = np.full(hdb.n, 0xFFFFFFFF, dtype=np.uint32)
    heavy_variants = candidates = 0
    for shard in range(nshards):
        assert gpu_ctx.d1_index_build() is False           # the fastidious pass re-purposes the table
        g, c = gpu_ctx.d1_fastidious(flags, stats[2], 16, shard, nshards)
        assert int(c[0]) == int(counters[0]) and (int(c[3]), int(c[4])) == (int(counters[3]), int(counters[4]))
        merged = np.minimum(merged, g)
        heavy_variants += int(c[1])
        candidates += int(c[2])
    assert np.array_equal(merged, graft)
    assert (heavy_variants, candidates) == (int(counters[1]), int(counters[2]))
