"""Test support: FASTA -> db arrays (numpy), ctypes bindings for the oracle and for the
reference shim, synthetic input generation.

Nothing in here is product code.  The db construction below is an independent numpy
restatement of the reference's ``db_read`` ordering and packing
(/root/reference/src/db.cc:100-114, 161-211, 388-413, 541-628) so that it can also
cross-check the product's own FASTA reader.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
REF_DIR = ORACLE_DIR / "_ref"
GOLDEN = ROOT / "tests" / "golden"

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------- db

@dataclass
class Db:
    headers: list            # bytes, in db order
    seqs: np.ndarray         # u64 words
    seq_off: np.ndarray      # u64 [n+1]
    seqlen: np.ndarray       # u32 [n]
    abundance: np.ndarray    # u64 [n]
    longest: int

    @property
    def n(self) -> int:
        return int(self.seqlen.shape[0])

    def words(self, i: int) -> np.ndarray:
        return self.seqs[int(self.seq_off[i]):int(self.seq_off[i + 1])]

    def seq_str(self, i: int) -> str:
        w = self.words(i)
        return "".join("ACGT"[(int(w[p >> 5]) >> ((p & 31) * 2)) & 3] for p in range(int(self.seqlen[i])))


_NT = np.full(256, 255, dtype=np.uint8)
for _c, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
    _NT[ord(_c)] = _v
    _NT[ord(_c.lower())] = _v


def pack_seq(seq: bytes) -> np.ndarray:
    codes = _NT[np.frombuffer(seq, dtype=np.uint8)]
    if (codes == 255).any():
        raise ValueError("illegal character in sequence")
    n = len(codes)
    nw = (n + 31) // 32
    padded = np.zeros(nw * 32, dtype=np.uint64)
    padded[:n] = codes
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))
    return (padded.reshape(nw, 32) << shifts).sum(axis=1, dtype=np.uint64)


def read_fasta(path) -> list:
    """[(header_bytes_up_to_first_space, sequence_bytes)] in file order."""
    recs = []
    hdr = None
    chunks = []
    with open(path, "rb") as fh:
        for line in fh:
            if line.startswith(b">"):
                if hdr is not None:
                    recs.append((hdr, b"".join(chunks)))
                hdr = re.split(rb"[ \r\n]", line[1:], maxsplit=1)[0]
                chunks = []
            else:
                chunks.append(line.strip())
    if hdr is not None:
        recs.append((hdr, b"".join(chunks)))
    return recs


_AB_SWARM = re.compile(rb"_([0-9]+)$")
_AB_USEARCH = re.compile(rb"(?:^|;)size=([0-9]+)(?:;|$)")


def abundance_of(header: bytes, usearch: bool = False) -> int:
    m = (_AB_USEARCH.search(header) if usearch else _AB_SWARM.search(header))
    if not m:
        raise ValueError(f"no abundance in {header!r}")
    return int(m.group(1))


def build_db(records, usearch: bool = False) -> Db:
    """Sort by (abundance desc, header bytes asc) and 2-bit pack."""
    recs = [(abundance_of(h, usearch), h, s) for h, s in records]
    recs.sort(key=lambda r: (-r[0], r[1]))
    packed = [pack_seq(s) for _, _, s in recs]
    seqlen = np.array([len(s) for _, _, s in recs], dtype=np.uint32)
    off = np.zeros(len(recs) + 1, dtype=np.uint64)
    if recs:
        off[1:] = np.cumsum([len(p) for p in packed], dtype=np.uint64)
    seqs = np.concatenate(packed) if packed else np.zeros(1, dtype=np.uint64)
    return Db(headers=[h for _, h, _ in recs], seqs=np.ascontiguousarray(seqs, dtype=np.uint64),
              seq_off=off, seqlen=seqlen,
              abundance=np.array([a for a, _, _ in recs], dtype=np.uint64),
              longest=int(seqlen.max()) if len(recs) else 0)


def db_from_fasta(path, usearch: bool = False) -> Db:
    return build_db(read_fasta(path), usearch)


# ----------------------------------------------------------------------- oracle

class OrcDb(C.Structure):
    _fields_ = [("n", C.c_uint32), ("longest", C.c_uint32), ("seqs", u64p), ("seq_off", u64p),
                ("seqlen", u32p), ("abundance", u64p)]


class OrcVar(C.Structure):
    _fields_ = [("hash", C.c_uint64), ("pos", C.c_uint32), ("type", C.c_uint8), ("base", C.c_uint8),
                ("pad", C.c_uint16)]


def _make(target: str) -> None:
    subprocess.run(["make", "-C", str(ORACLE_DIR), target], check=True, stdout=subprocess.DEVNULL)


_oracle = None


def oracle():
    """ctypes handle of oracle/liboracle.so (built on demand)."""
    global _oracle
    if _oracle is None:
        _make("all")
        lib = C.CDLL(str(ORACLE_DIR / "liboracle.so"))
        lib.orc_zobrist_table.argtypes = [C.c_uint32, u64p]
        lib.orc_zobrist_hash.restype = C.c_uint64
        lib.orc_zobrist_hash.argtypes = [u64p, u64p, C.c_uint32]
        lib.orc_zobrist_hash_delete_first.restype = C.c_uint64
        lib.orc_zobrist_hash_delete_first.argtypes = [u64p, u64p, C.c_uint32]
        lib.orc_zobrist_hash_insert_first.restype = C.c_uint64
        lib.orc_zobrist_hash_insert_first.argtypes = [u64p, u64p, C.c_uint32]
        lib.orc_generate_variants.restype = C.c_uint32
        lib.orc_generate_variants.argtypes = [u64p, u64p, C.c_uint32, C.c_uint64, C.POINTER(OrcVar)]
        lib.orc_check_variant.restype = C.c_int
        lib.orc_check_variant.argtypes = [u64p, C.c_uint32, C.POINTER(OrcVar), u64p, C.c_uint32]
        lib.orc_generate_variant_sequence.restype = C.c_uint32
        lib.orc_generate_variant_sequence.argtypes = [u64p, C.c_uint32, C.POINTER(OrcVar), u64p]
        lib.orc_hashtable_size.restype = C.c_uint64
        lib.orc_hashtable_size.argtypes = [C.c_uint64]
        lib.orc_bloom_patterns.argtypes = [u64p]
        lib.orc_bloomflex_patterns.argtypes = [C.c_uint32, u64p]
        lib.orc_d1_index_build.restype = C.c_void_p
        lib.orc_d1_index_build.argtypes = [C.POINTER(OrcDb), C.POINTER(C.c_int)]
        lib.orc_d1_index_free.argtypes = [C.c_void_p]
        lib.orc_d1_network.restype = C.c_uint64
        lib.orc_d1_network.argtypes = [C.POINTER(OrcDb), C.c_void_p, C.c_int, C.c_uint32, C.c_uint32,
                                       u64p, u32p, C.c_uint64]
        lib.orc_d1_fastidious.restype = C.c_int
        lib.orc_d1_fastidious.argtypes = [C.POINTER(OrcDb), u8p, C.c_uint64, C.c_uint32, u32p, u64p]
        lib.orc_findqgrams.argtypes = [u64p, C.c_uint32, u8p]
        lib.orc_qgram_diff.restype = C.c_uint64
        lib.orc_qgram_diff.argtypes = [u8p, u8p]
        lib.orc_nw_diff.restype = C.c_uint64
        lib.orc_nw_diff.argtypes = [u64p, C.c_uint32, u64p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                                    u64p, u64p]
        _oracle = lib
    return _oracle


def orc_db(db: Db) -> OrcDb:
    o = OrcDb(db.n, db.longest, _p(db.seqs, u64p), _p(db.seq_off, u64p), _p(db.seqlen, u32p),
              _p(db.abundance, u64p))
    o._keep = db  # keep arrays alive
    return o


def oracle_zobrist(zlen: int) -> np.ndarray:
    tab = np.zeros(4 * zlen, dtype=np.uint64)
    oracle().orc_zobrist_table(zlen, _p(tab, u64p))
    return tab


def oracle_variants(tab: np.ndarray, words: np.ndarray, length: int, h: int):
    out = (OrcVar * (7 * length + 5))()
    n = oracle().orc_generate_variants(_p(tab, u64p), _p(words, u64p), length, h, out)
    return [(out[i].hash, out[i].pos, out[i].type, out[i].base) for i in range(n)]


def oracle_d1_network(db: Db, no_cluster_breaking: bool = False, first: int = 0, count: int | None = None):
    """CSR (offsets u64[count+1], neighbours u32[total]) with rows in the reference's hit order."""
    lib = oracle()
    odb = orc_db(db)
    dup = C.c_int(0)
    ix = lib.orc_d1_index_build(C.byref(odb), C.byref(dup))
    assert ix
    try:
        if count is None:
            count = db.n - first
        offsets = np.zeros(count + 1, dtype=np.uint64)
        cap = max(16, 16 * count)
        while True:
            nb = np.zeros(cap, dtype=np.uint32)
            total = lib.orc_d1_network(C.byref(odb), ix, int(no_cluster_breaking), first, count,
                                       _p(offsets, u64p), _p(nb, u32p), cap)
            if total <= cap:
                return offsets, nb[:total], bool(dup.value)
            cap = int(total)
    finally:
        lib.orc_d1_index_free(ix)


class OrcD1IndexHead(C.Structure):
    """the leading fields of orc_d1_index (oracle/swarm_oracle.h): enough to read the Bloom bitmap"""
    _fields_ = [("table_size", C.c_uint64), ("hash_values", C.c_void_p), ("hash_data", C.c_void_p), ("hash_occupied", C.c_void_p),
                ("bloom", C.POINTER(C.c_uint64)), ("bloom_mask", C.c_uint64)]


def oracle_d1_bloom(db: Db) -> np.ndarray:
    """the blocked Bloom filter in front of the amplicon table as the oracle builds it (src/bloompat.cc; one bit per
    table slot x 8 = table_size / 8 64-bit words, at least one)"""
    lib = oracle()
    odb = orc_db(db)
    dup = C.c_int(0)
    ix = lib.orc_d1_index_build(C.byref(odb), C.byref(dup))
    assert ix
    try:
        head = C.cast(ix, C.POINTER(OrcD1IndexHead)).contents
        words = int(head.bloom_mask) + 1
        return np.ctypeslib.as_array(head.bloom, shape=(words,)).copy()
    finally:
        lib.orc_d1_index_free(ix)


def oracle_derep(db: Db) -> np.ndarray:
    lib = oracle()
    lib.orc_derep.restype = C.c_int
    lib.orc_derep.argtypes = [C.POINTER(OrcDb), u32p]
    odb = orc_db(db)
    out = np.zeros(db.n, dtype=np.uint32)
    assert lib.orc_derep(C.byref(odb), _p(out, u32p)) == 0
    return out


def oracle_fastidious(db: Db, is_light: np.ndarray, bloom_bits: int = 16):
    lib = oracle()
    odb = orc_db(db)
    is_light = np.ascontiguousarray(is_light, dtype=np.uint8)
    light_nt = int(db.seqlen[is_light != 0].astype(np.uint64).sum())
    graft = np.zeros(db.n, dtype=np.uint32)
    counters = np.zeros(8, dtype=np.uint64)
    rc = lib.orc_d1_fastidious(C.byref(odb), _p(is_light, u8p), light_nt, bloom_bits, _p(graft, u32p),
                               _p(counters, u64p))
    assert rc == 0
    return graft, counters


# -------------------------------------------------------------------- reference

def have_reference() -> bool:
    """True when the compiled reference (oracle/_ref) is available or can be built."""
    if (REF_DIR / "swarm").exists() and (REF_DIR / "libswarmref.so").exists():
        return True
    if Path("/root/reference/src/swarm.cc").exists():
        try:
            _make("ref")
        except Exception:
            return False
        return (REF_DIR / "swarm").exists()
    return False


def ref_swarm_bin() -> Path:
    return REF_DIR / "swarm"


def run_ref_swarm(args: list, cwd=None) -> subprocess.CompletedProcess:
    return subprocess.run([str(ref_swarm_bin())] + [str(a) for a in args], cwd=cwd,
                          capture_output=True, text=True)


# -------------------------------------------------------------------- generator

def gen_bin() -> Path:
    out = ROOT / "tools" / "gen_amplicons"
    src = ROOT / "tools" / "gen_amplicons.c"
    if not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-o", str(out), str(src), "-lm"], check=True)
    return out


def gen_fasta(path, n: int, length: int, seed: int, max_edits: int = 1, light_frac: float = 0.0, env: dict | None = None) -> None:
    """tools/gen_amplicons; env: the generator's switches (GEN_FLANK, GEN_CORE, GEN_ZIPF, GEN_CONSERVED)"""
    subprocess.run([str(gen_bin()), str(n), str(length), str(seed), str(max_edits), str(light_frac), str(path)],
                   check=True, env=dict(os.environ, **env) if env else None)


# -------------------------------------------------------------------- device memory poison

def poison_device_memory(chunks: int = 32, chunk_bytes: int = 1 << 28) -> None:
    """Fill `chunks` x `chunk_bytes` of HBM with non-zero bytes and free it again, through the same HIP runtime the
    library uses (libamdhip64.so, already loaded with it): what the library allocates next has been 0xA5 / 0xFF /
    0x01 / 0x5A, not zero.  (torch is deliberately not used here: its wheel carries its own copy of the HIP
    runtime, and initialising that one after the library's has been seen to fail with "No HIP GPUs are available".)"""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    held = []
    for k in range(chunks):
        ptr = C.c_void_p()
        if hip.hipMalloc(C.byref(ptr), chunk_bytes) != 0:
            break
        assert hip.hipMemset(ptr, (0xA5, 0xFF, 0x01, 0x5A)[k % 4], chunk_bytes) == 0
        held.append(ptr)
    assert hip.hipDeviceSynchronize() == 0 and len(held) > 0
    for ptr in held:
        assert hip.hipFree(ptr) == 0


class DeviceArray:
    """A plain device buffer through the HIP runtime the library uses (no torch): what the device-pointer entry
    points of the C ABI take in the tests.  data_ptr() like a torch tensor."""

    def __init__(self, count: int, dtype=np.uint32):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.dtype, self.count = np.dtype(dtype), int(count)
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), max(1, self.count) * self.dtype.itemsize) == 0

    def data_ptr(self) -> int:
        return self.ptr.value

    def numel(self) -> int:
        return self.count

    def to_host(self, count: int | None = None) -> np.ndarray:
        out = np.empty(self.count if count is None else count, dtype=self.dtype)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2) == 0      # hipMemcpyDeviceToHost
        return out

    def from_host(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size <= self.count
        assert self.hip.hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0           # hipMemcpyHostToDevice

    def free(self) -> None:
        if self.ptr:
            self.hip.hipFree(self.ptr)
            self.ptr = C.c_void_p()
