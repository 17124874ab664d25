"""The oracle (oracle/swarm_oracle.c) against the committed golden fixtures that were produced
by the unmodified reference (tests/golden/make_golden.py).  CPU only."""
import ctypes as C
import json
import re

import numpy as np
import pytest

import support as S

G = S.GOLDEN
VEC = json.loads((G / "function_vectors.json").read_text())


def test_zobrist_table_matches_reference():
    tab = S.oracle_zobrist(450)
    assert [int(x) for x in tab[:32]] == VEC["zobrist_first_32"]
    assert int(np.bitwise_xor.reduce(tab)) == VEC["zobrist_xor_all_1800"]


def test_bloom_patterns_match_reference():
    lib = S.oracle()
    pat = np.zeros(1024, dtype=np.uint64)
    lib.orc_bloom_patterns(S._p(pat, S.u64p))
    assert [int(x) for x in pat[:8]] == VEC["bloom_patterns_first_8"]
    assert int(np.bitwise_xor.reduce(pat)) == VEC["bloom_patterns_xor"]
    assert all(bin(int(x)).count("1") == 8 for x in pat)
    fp = np.zeros(65536, dtype=np.uint64)
    lib.orc_bloomflex_patterns(6, S._p(fp, S.u64p))
    assert [int(x) for x in fp[:8]] == VEC["bloomflex_k6_first_8"]
    assert int(np.bitwise_xor.reduce(fp)) == VEC["bloomflex_k6_xor"]


def test_hashtable_size_table():
    lib = S.oracle()
    for n, want in VEC["hashtable_size"].items():
        assert lib.orc_hashtable_size(int(n)) == want, n


@pytest.mark.parametrize("rec", VEC["sequences"], ids=lambda r: f"L{len(r['seq'])}")
def test_hashes_variants_qgrams(rec):
    lib = S.oracle()
    tab = S.oracle_zobrist(450)
    s = rec["seq"]
    w = S.pack_seq(s.encode())
    tp, wp = S._p(tab, S.u64p), S._p(w, S.u64p)
    h = lib.orc_zobrist_hash(tp, wp, len(s))
    assert h == rec["hash"]
    assert lib.orc_zobrist_hash_delete_first(tp, wp, len(s)) == rec["hash_delete_first"]
    assert lib.orc_zobrist_hash_insert_first(tp, wp, len(s)) == rec["hash_insert_first"]
    got = S.oracle_variants(tab, w, len(s), h)
    assert len(got) == rec["n_variants"]
    x = 0
    for v in got:
        x ^= v[0]
    assert x == rec["variant_hash_xor"]
    if rec["variants"] is not None:
        assert [list(v) for v in got] == rec["variants"]
    q = np.zeros(128, dtype=np.uint8)
    lib.orc_findqgrams(wp, len(s), S._p(q, S.u8p))
    assert q.tobytes().hex() == rec["qgram_hex"]


def test_variants_roundtrip_check_variant():
    """every generated variant, materialised, verifies against itself and has the right hash"""
    lib = S.oracle()
    tab = S.oracle_zobrist(450)
    for rec in VEC["sequences"]:
        s = rec["seq"]
        if len(s) > 160:
            continue
        w = S.pack_seq(s.encode())
        out = (S.OrcVar * (7 * len(s) + 5))()
        n = lib.orc_generate_variants(S._p(tab, S.u64p), S._p(w, S.u64p), len(s), rec["hash"], out)
        seen = set()
        for i in range(n):
            buf = np.zeros(len(w) + 2, dtype=np.uint64)
            vlen = lib.orc_generate_variant_sequence(S._p(w, S.u64p), len(s), C.byref(out[i]), S._p(buf, S.u64p))
            if vlen == 0:
                assert len(s) == 1 and out[i].type == 1
                continue
            assert lib.orc_zobrist_hash(S._p(tab, S.u64p), S._p(buf, S.u64p), vlen) == out[i].hash
            assert lib.orc_check_variant(S._p(w, S.u64p), len(s), C.byref(out[i]), S._p(buf, S.u64p), vlen) == 1
            key = (vlen, buf.tobytes())
            assert key not in seen          # variants are distinct sequences
            seen.add(key)


def test_nw_known_answers():
    lib = S.oracle()
    for p in VEC["nw_pairs"]:
        q = S.pack_seq(p["q"].encode())
        d = S.pack_seq(p["d"].encode())
        alen = C.c_uint64(0)
        got = lib.orc_nw_diff(S._p(d, S.u64p), len(p["d"]), S._p(q, S.u64p), len(p["q"]), p["mismatch"], p["gapopen"],
                              p["gapextend"], C.byref(alen), None)
        assert (got, alen.value) == (p["diff"], p["alnlen"]), p


@pytest.mark.parametrize("name", ["d1_1k", "d1_nobreak", "d1_short"])
def test_network_matches_reference_network_file(name):
    db = S.db_from_fasta(G / f"{name}.fasta")
    ncb = "-n" in (G / f"{name}.args").read_text().split()
    off, nb, dup = S.oracle_d1_network(db, ncb)
    assert not dup
    lines = []
    for i in range(db.n):
        for j in sorted(nb[int(off[i]):int(off[i + 1])].tolist()):
            lines.append(db.headers[i] + b"\t" + db.headers[j] + b"\n")
    assert b"".join(lines) == (G / f"{name}.j").read_bytes()


def _log_numbers(name):
    log = (G / f"{name}.log").read_text()
    pick = lambda pat: int(re.search(pat, log).group(1))
    return {"light_variants": pick(r"Generated (\d+) variants"), "heavy_variants": pick(r"Heavy variants: (\d+)"),
            "candidates": pick(r"Got (\d+) graft"), "grafts": pick(r"Made (\d+) grafts"), "m": pick(r"m=(\d+)"),
            "k": pick(r"k=(\d+)"), "light_nt": pick(r"light swarms: (\d+)"),
            "light_swarms": pick(r"Light swarms: (\d+)"), "heavy_swarms": pick(r"Heavy swarms: (\d+)")}


def _simple_clusters(db, off, nb):
    """plain-Python restatement of the greedy d=1 walk (src/algod1.cc:1185-1280) for small cases"""
    swarm = [-1] * db.n
    masses = []
    for seed in range(db.n):
        if swarm[seed] >= 0:
            continue
        sid = len(masses)
        swarm[seed] = sid
        frontier, mass = [seed], int(db.abundance[seed])
        while frontier:
            fresh = []
            for s in frontier:
                for a in nb[int(off[s]):int(off[s + 1])].tolist():
                    if swarm[a] < 0:
                        swarm[a] = sid
                        mass += int(db.abundance[a])
                        fresh.append(a)
            frontier = sorted(fresh)
        masses.append(mass)
    return swarm, masses


@pytest.mark.parametrize("name,boundary,bits", [("d1_fastidious", 3, 16), ("d1_fastidious_b10_y8", 10, 8)])
def test_fastidious_counters_match_reference_log(name, boundary, bits):
    db = S.db_from_fasta(G / f"{name}.fasta")
    off, nb, _ = S.oracle_d1_network(db)
    swarm, masses = _simple_clusters(db, off, nb)
    is_light = np.array([1 if masses[swarm[i]] < boundary else 0 for i in range(db.n)], dtype=np.uint8)
    want = _log_numbers(name)
    assert int(db.seqlen[is_light != 0].astype(np.uint64).sum()) == want["light_nt"]
    graft, counters = S.oracle_fastidious(db, is_light, bits)
    assert int(counters[0]) == want["light_variants"]
    assert int(counters[1]) == want["heavy_variants"]
    assert int(counters[2]) == want["candidates"]
    assert (int(counters[3]), int(counters[4])) == (want["m"], want["k"])
    # graft candidates only on light amplicons, always pointing at heavy ones
    has = graft != 0xFFFFFFFF
    assert (is_light[has] == 1).all() and (is_light[graft[has].astype(np.int64)] == 0).all()
