"""B3 (q-gram prefilter) and B4 (alignment scan) parity on the GPU, through the C ABI, against
the oracle (orc_findqgrams / orc_qgram_diff / orc_nw_diff) and the committed golden vectors."""
import ctypes as C
import json

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu
G = S.GOLDEN


def _upload(ctx, db):
    ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)


def _oracle_sig(db, i):
    q = np.zeros(128, dtype=np.uint8)
    S.oracle().orc_findqgrams(S._p(db.words(i), S.u64p), int(db.seqlen[i]), S._p(q, S.u8p))
    return q


def _oracle_nw(db, q, t, mm, go, ge):
    alen = C.c_uint64(0)
    score = C.c_uint64(0)
    d = S.oracle().orc_nw_diff(S._p(db.words(t), S.u64p), int(db.seqlen[t]), S._p(db.words(q), S.u64p),
                               int(db.seqlen[q]), mm, go, ge, C.byref(alen), C.byref(score))
    return int(d), int(alen.value), int(score.value)


@pytest.mark.parametrize("n,L,seed,edits", [(600, 120, 51, 3), (300, 400, 52, 3), (200, 33, 53, 2), (150, 7, 54, 1),
                                            (64, 4, 55, 1)])
def test_qgram_signatures_and_diffs(gpu_ctx, tmp_path, n, L, seed, edits):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, max(L, 8), seed, edits)
    recs = S.read_fasta(fa)
    if L < 8:                                   # shorter than the generator allows: truncate, keep unique
        seen, out = set(), []
        for h, s in recs:
            s = s[:L + (len(out) % 3)]
            if s not in seen:
                seen.add(s)
                out.append((h, s))
        recs = out
    db = S.build_db(recs)
    _upload(gpu_ctx, db)
    gpu_ctx.qgram_build()
    sigs = gpu_ctx.qgram_signatures()
    want = np.stack([_oracle_sig(db, i) for i in range(db.n)])
    assert np.array_equal(sigs, want)
    rng = np.random.default_rng(seed)
    lib = S.oracle()
    for seed_amp in rng.integers(0, db.n, size=4):
        for listlen in (1, 7, 255, 256, db.n):
            lst = rng.integers(0, db.n, size=min(listlen, db.n)).astype(np.uint64)
            got = gpu_ctx.qgram_diff(int(seed_amp), lst)
            exp = np.array([lib.orc_qgram_diff(S._p(want[int(seed_amp)], S.u8p), S._p(want[int(t)], S.u8p)) for t in lst],
                           dtype=np.uint64)
            assert np.array_equal(got, exp)
    assert len(gpu_ctx.qgram_diff(0, np.zeros(0, dtype=np.uint64))) == 0


def test_qgram_golden_vectors(gpu_ctx):
    vec = json.loads((G / "function_vectors.json").read_text())["sequences"]
    db = S.build_db([(f"s{i}_{len(vec) - i}".encode(), r["seq"].encode()) for i, r in enumerate(vec)])
    _upload(gpu_ctx, db)
    gpu_ctx.qgram_build()
    sigs = gpu_ctx.qgram_signatures()
    for i, r in enumerate(vec):
        assert sigs[i].tobytes().hex() == r["qgram_hex"]


def _unique_costs(mm, go, ge, d):
    T = d * max(mm, go + ge)
    ways = {}
    a = 0
    while a * mm <= T:
        g = 0
        while a * mm + g * (go + ge) <= T:
            e = g
            while a * mm + g * go + e * ge <= T:
                ways.setdefault(a * mm + g * go + e * ge, set()).add((a + e, e))
                if g == 0:
                    break
                e += 1
            g += 1
        a += 1
    return all(len(v) == 1 for v in ways.values()) and len(ways) <= 64


SCORINGS = [((18, 24, 13), 3), ((18, 24, 13), 1), ((18, 24, 13), 2), ((18, 24, 13), 4), ((18, 24, 13), 5), ((18, 24, 13), 6),
            ((18, 24, 13), 7), ((4, 12, 1), 2), ((2, 3, 1), 5), ((18, 24, 13), 9), ((7, 11, 3), 3), ((10, 1, 10), 2)]


@pytest.mark.parametrize("scoring,d", SCORINGS)
@pytest.mark.parametrize("alphabet", ["ACGT", "AC"])
def test_alignment_diffs_match_oracle(gpu_ctx, scoring, d, alphabet):
    """accepted pairs (oracle diff <= d) must be bit-identical; rejected pairs must stay > d"""
    mm, go, ge = scoring
    rng = np.random.default_rng(d * 100 + mm + len(alphabet))
    recs = []
    for k in range(40):
        L = int(rng.integers(8, 160))
        base = "".join(rng.choice(list(alphabet), size=L))
        recs.append(base)
        for _ in range(9):
            s = list(base)
            for _e in range(int(rng.integers(0, d + 3))):
                u = rng.random()
                p = int(rng.integers(0, len(s)))
                if u < 0.5:
                    s[p] = alphabet[int(rng.integers(0, len(alphabet)))]
                elif u < 0.75 and len(s) > 4:
                    del s[p]
                else:
                    s.insert(p, alphabet[int(rng.integers(0, len(alphabet)))])
            recs.append("".join(s))
    recs = sorted(set(recs))
    db = S.build_db([(f"s{i}_{1 + (i * 7) % 50}".encode(), s.encode()) for i, s in enumerate(recs)])
    _upload(gpu_ctx, db)
    gpu_ctx.search_begin(mm, go, ge, d)
    # the wavefront kernel is used exactly when every cost <= T has one decomposition (default
    # penalties: d <= 3) and the band fits a 32-lane group; both kernels must pass the same checks
    assert gpu_ctx.search_uses_wavefront() == _unique_costs(mm, go, ge, d)
    sat = 65535 if d > min(255 // mm, 255 // (go + ge)) else 255
    accepted = 0
    for q in rng.integers(0, db.n, size=12):
        targets = np.array([t for t in range(db.n) if t != q], dtype=np.uint64)
        scores, diffs, alens = gpu_ctx.search_do(int(q), targets)
        for k, t in enumerate(targets):
            want, walen, wscore = _oracle_nw(db, int(q), int(t), mm, go, ge)
            if want <= d and wscore < sat:
                accepted += 1
                assert (int(diffs[k]), int(alens[k]), int(scores[k])) == (want, walen, wscore), (q, t)
            else:
                assert int(diffs[k]) > d, (q, t, want, int(diffs[k]))
    assert accepted > 2


def test_alignment_golden_vectors(gpu_ctx):
    pairs = json.loads((G / "function_vectors.json").read_text())["nw_pairs"]
    by_scoring = {}
    for p in pairs:
        by_scoring.setdefault((p["mismatch"], p["gapopen"], p["gapextend"]), []).append(p)
    for (mm, go, ge), ps in by_scoring.items():
        seqs = sorted({p["q"] for p in ps} | {p["d"] for p in ps})
        index = {s: i for i, s in enumerate(seqs)}
        # abundance descending keeps the list order as db order
        db = S.build_db([(f"s{i:05d}_{len(seqs) - i}".encode(), s.encode()) for i, s in enumerate(seqs)])
        assert [db.seq_str(i) for i in range(db.n)] == seqs
        _upload(gpu_ctx, db)
        d = 6
        gpu_ctx.search_begin(mm, go, ge, d)
        for p in ps:
            scores, diffs, alens = gpu_ctx.search_do(index[p["q"]], np.array([index[p["d"]]], dtype=np.uint64))
            if p["diff"] <= d:
                assert (int(diffs[0]), int(alens[0])) == (p["diff"], p["alnlen"]), p
            else:
                assert int(diffs[0]) > d, p


@pytest.mark.parametrize("d", [1, 2, 3])
def test_wavefront_and_banded_kernels_agree(gpu_ctx, d):
    """Same pairs through both B4 kernels (SWA_ALIGN_BANDED forces the banded one): wherever
    either accepts (diff <= d) the (diff, length, score) triples are identical, otherwise both
    reject.  Long sequences, mixed lengths, many near-identical pairs."""
    import os
    rng = np.random.default_rng(700 + d)
    recs = set()
    for _ in range(25):
        L = int(rng.integers(150, 420))
        base = "".join(rng.choice(list("ACGT"), size=L))
        recs.add(base)
        for _v in range(12):
            s = list(base)
            for _e in range(int(rng.integers(0, d + 3))):
                u = rng.random()
                p = int(rng.integers(0, len(s)))
                if u < 0.5:
                    s[p] = "ACGT"[int(rng.integers(0, 4))]
                elif u < 0.75:
                    del s[p]
                else:
                    s.insert(p, "ACGT"[int(rng.integers(0, 4))])
            recs.add("".join(s))
    recs = sorted(recs)
    db = S.build_db([(f"s{i}_{1 + i % 9}".encode(), s.encode()) for i, s in enumerate(recs)])
    _upload(gpu_ctx, db)
    gpu_ctx.search_begin(18, 24, 13, d)
    assert gpu_ctx.search_uses_wavefront()
    accepted = 0
    try:
        for q in rng.integers(0, db.n, size=10):
            targets = np.array([t for t in range(db.n) if t != q], dtype=np.uint64)
            os.environ.pop("SWA_ALIGN_BANDED", None)
            ws, wd, wl = gpu_ctx.search_do(int(q), targets)
            os.environ["SWA_ALIGN_BANDED"] = "1"
            bs, bd, bl = gpu_ctx.search_do(int(q), targets)
            acc = (bd <= d) | (wd <= d)
            accepted += int(acc.sum())
            assert np.array_equal(wd[acc], bd[acc]) and np.array_equal(wl[acc], bl[acc]) and np.array_equal(ws[acc], bs[acc])
            assert (wd[~acc] > d).all() and (bd[~acc] > d).all()
    finally:
        os.environ.pop("SWA_ALIGN_BANDED", None)
    assert accepted >= 3
