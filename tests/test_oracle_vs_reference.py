"""Fuzz the oracle against the compiled, unmodified reference (oracle/_ref, built from
/root/reference by oracle/Makefile).  Skipped where the reference is not available (GPU box
runs only the golden-fixture version of these checks)."""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.skipif(not S.have_reference(), reason="compiled reference (oracle/_ref) not available")

_FUZZ = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
import support as S
lib = S.oracle()
ref = C.CDLL(str(S.REF_DIR / "libswarmref.so"))
for f in ("ref_zobrist_table", "ref_zobrist_hash", "ref_zobrist_hash_delete_first", "ref_zobrist_hash_insert_first", "ref_nw"):
    getattr(ref, f).restype = C.c_uint64
ZL = 700
assert ref.ref_zobrist_init(ZL) == 0
rt = np.zeros(4 * ZL, dtype=np.uint64)
ref.ref_zobrist_table(rt.ctypes.data_as(S.u64p), C.c_uint64(4 * ZL))
ot = S.oracle_zobrist(ZL)
assert np.array_equal(rt, ot)
rng = np.random.default_rng(int(sys.argv[1]))
for trial in range(250):
    L = int(rng.integers(1, 600))
    alpha = [b"ACGT", b"AC", b"A", b"ACGT"][trial %% 4]
    s = bytes(rng.choice(list(alpha), size=L).tolist())
    w = S.pack_seq(s)
    wp = w.ctypes.data_as(S.u64p); cp = w.ctypes.data_as(C.c_char_p); tp = ot.ctypes.data_as(S.u64p)
    h = lib.orc_zobrist_hash(tp, wp, L)
    assert h == ref.ref_zobrist_hash(cp, L)
    assert lib.orc_zobrist_hash_delete_first(tp, wp, L) == ref.ref_zobrist_hash_delete_first(cp, L)
    assert lib.orc_zobrist_hash_insert_first(tp, wp, L) == ref.ref_zobrist_hash_insert_first(cp, L)
    N = 7 * L + 5
    oh = np.zeros(N, dtype=np.uint64); op = np.zeros(N, dtype=np.uint32); oty = np.zeros(N, dtype=np.uint8); ob = np.zeros(N, dtype=np.uint8)
    n = ref.ref_generate_variants(cp, L, C.c_uint64(h), oh.ctypes.data_as(S.u64p), op.ctypes.data_as(S.u32p),
                                  oty.ctypes.data_as(S.u8p), ob.ctypes.data_as(S.u8p))
    assert [(int(oh[i]), int(op[i]), int(oty[i]), int(ob[i])) for i in range(n)] == S.oracle_variants(ot, w, L, h)
    qa = np.zeros(128, dtype=np.uint8); qb = np.zeros(128, dtype=np.uint8)
    ref.ref_findqgrams(cp, C.c_uint64(L), qa.ctypes.data_as(S.u8p))
    lib.orc_findqgrams(wp, L, qb.ctypes.data_as(S.u8p))
    assert np.array_equal(qa, qb)
for trial in range(300):
    L = int(rng.integers(1, 120))
    alpha = list("ACGT" if trial %% 2 else "AC")
    a = "".join(rng.choice(alpha, size=L))
    b = list(a)
    for _ in range(int(rng.integers(0, 6))):
        p = int(rng.integers(0, len(b))) if b else 0
        u = rng.random()
        if u < 0.5 and b: b[p] = alpha[int(rng.integers(0, len(alpha)))]
        elif u < 0.75 and len(b) > 1: del b[p]
        else: b.insert(p, alpha[int(rng.integers(0, len(alpha)))])
    b = "".join(b)
    wa = S.pack_seq(a.encode()); wb = S.pack_seq(b.encode())
    mm, go, ge = [(18, 24, 13), (4, 12, 1), (2, 3, 1)][trial %% 3]
    al = C.c_uint64(0); buf = C.create_string_buffer(len(a) + len(b) + 8)
    want = ref.ref_nw(wb.ctypes.data_as(C.c_char_p), C.c_uint64(len(b)), wa.ctypes.data_as(C.c_char_p), C.c_uint64(len(a)),
                      C.c_int64(mm), C.c_uint64(go), C.c_uint64(ge), buf, C.byref(al))
    al2 = C.c_uint64(0)
    got = lib.orc_nw_diff(wb.ctypes.data_as(S.u64p), len(b), wa.ctypes.data_as(S.u64p), len(a), mm, go, ge, C.byref(al2), None)
    assert (got, al2.value) == (want, al.value), (a, b, mm, go, ge)
print("ok")
'''


@pytest.mark.parametrize("seed", [1, 2])
def test_function_level_fuzz(seed):
    # own process: the reference's generators are only valid on their first call per process
    r = subprocess.run([sys.executable, "-c", _FUZZ % str(S.ROOT / "tests"), str(seed)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


@pytest.mark.parametrize("n,length,seed,ncb", [(2500, 150, 101, False), (1500, 64, 102, True), (800, 300, 103, False)])
def test_network_against_reference_binary(tmp_path, n, length, seed, ncb):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, length, seed)
    net = tmp_path / "net.txt"
    r = S.run_ref_swarm(["-d", "1", "-j", net, "-o", "/dev/null", "-l", "/dev/null"] + (["-n"] if ncb else []) + [fa])
    assert r.returncode == 0, r.stderr
    db = S.db_from_fasta(fa)
    off, nb, _ = S.oracle_d1_network(db, ncb)
    lines = []
    for i in range(db.n):
        for j in sorted(nb[int(off[i]):int(off[i + 1])].tolist()):
            lines.append(db.headers[i] + b"\t" + db.headers[j] + b"\n")
    assert b"".join(lines) == net.read_bytes()
