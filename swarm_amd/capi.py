"""ctypes binding of libswarm_amd.so (include/swarm_amd.h) — the thin Python face of the
C ABI used by the tests, bench.py and __graft_entry__.py.

There is deliberately NO fallback here: if the HIP library is missing or no gfx950
device is usable, everything raises.  Nothing in this package imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

# libgomp's default (workers spin between parallel regions) costs the host phases up to 30x on a many-core GPU host
# (see host/main.cpp); it reads the variable when it is first loaded, which importing this module usually precedes.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def usable_cpus() -> int:
    """CPUs this process may really use: affinity, cut down to the cgroup's CPU quota (what swa_host_cpus, host/pool.h,
    computes for the library's own worker threads) — a container that sees 256 CPUs and may use 16 is throttled as soon
    as more than 16 threads are busy."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, round(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, round(q / p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


# (The library sizes its own OpenMP teams — num_threads(swa_host_team()) on every parallel region, host/out.h — so this
# module does not touch OMP_NUM_THREADS: other OpenMP users of the importing process keep their own setting.)
import subprocess
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
# SWARM_AMD_LIB=<path>: another build of the library (triage: tools/gpu_selfcheck.py; `make asan`)
LIB_PATH = Path(os.environ.get("SWARM_AMD_LIB") or PKG / "lib" / "libswarm_amd.so")

SWA_OK, SWA_E_DEVICE, SWA_E_ARG, SWA_E_NOMEM, SWA_E_CAPACITY, SWA_E_DUPLICATES, SWA_E_INTERNAL = range(7)
NO_AMPLICON = 0xFFFFFFFF

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class SwaError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libswarm_amd error {code}: {msg}")
        self.code = code


class DbView(C.Structure):
    _fields_ = [("n", C.c_uint32), ("longest", C.c_uint32), ("seqs", C.c_void_p), ("seq_off", C.c_void_p),
                ("seqlen", C.c_void_p), ("abundance", C.c_void_p)]


class DbUnorderedView(C.Structure):
    """swa_db_unordered_view: the reader's word pools in file order + where every amplicon's words begin"""
    _fields_ = [("n", C.c_uint32), ("longest", C.c_uint32), ("pieces", C.c_uint32), ("piece_words", C.c_void_p),
                ("piece_word_count", C.c_void_p), ("src_off", C.c_void_p), ("seqlen", C.c_void_p), ("abundance", C.c_void_p)]


EXPORTS = [
    "swa_abi_version", "swa_ctx_create", "swa_ctx_destroy", "swa_last_error", "swa_ctx_synchronize", "swa_ctx_warmup", "swa_ctx_warmup_for", "swa_d1_anchor_windows", "swa_d1_anchor_width",
    "swa_d1_network_resident", "swa_d1_network_fetch", "swa_d1_cluster_device", "swa_d1_cluster_fetch", "swa_d1_cluster_maxgen", "swa_d1_cluster_resident", "swa_d1_cluster_resident_lazy", "swa_d1_result_detach", "swa_d1_result_error", "swa_d1_result_prepare",
    "swa_d1_cluster_resident_prepared", "swa_host_pin", "swa_host_unpin", "swa_ctx_warmup_downloads",
    "swa_db_upload", "swa_db_attach", "swa_db_stage_words", "swa_db_upload_unordered", "swa_hostdb_unordered_view", "swa_hostdb_read_fasta_staged", "swa_cli_main", "swa_d1_index_build", "swa_d1_index_build_range", "swa_d1_set_ownership", "swa_d1_route_slice", "swa_d1_index_build_routed", "swa_d1_route_slice_records", "swa_d1_index_build_records", "swa_d1_network", "swa_d1_network_edges_device", "swa_d1_network_device", "swa_d1_guard_retries",
    "swa_d1_debug_read", "swa_d1_table_size", "swa_search_uses_wavefront", "swa_d1_fastidious", "swa_d1_fastidious_shard", "swa_qgram_build", "swa_qgram_diff",
    "swa_qgram_debug_read", "swa_search_begin", "swa_search_do", "swa_timing_enable", "swa_timing_read",
    "swa_hostdb_read_fasta", "swa_hostdb_free", "swa_hostdb_error", "swa_hostdb_view", "swa_hostdb_nucleotides",
    "swa_hostdb_header", "swa_d1_cluster", "swa_d1_result_free", "swa_d1_result_summary", "swa_d1_result_swarmid",
    "swa_d1_result_parent", "swa_d1_result_generation", "swa_d1_light_flags", "swa_d1_graft", "swa_d1_write_swarms",
    "swa_d1_write_stats", "swa_d1_write_structure", "swa_d1_write_seeds", "swa_d1_write_network",
]


def build_library(force: bool = False) -> Path:
    """Compile every HIP translation unit for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", str(PKG / "csrc"), "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", str(PKG / "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load_library() -> C.CDLL:
    """dlopen libswarm_amd.so and declare the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
    lib = C.CDLL(str(LIB_PATH))
    lib.swa_abi_version.restype = C.c_int
    lib.swa_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.swa_ctx_destroy.argtypes = [C.c_void_p]
    lib.swa_ctx_destroy.restype = None
    lib.swa_last_error.argtypes = [C.c_void_p]
    lib.swa_last_error.restype = C.c_char_p
    lib.swa_ctx_synchronize.argtypes = [C.c_void_p]
    lib.swa_db_upload.argtypes = [C.c_void_p, C.POINTER(DbView)]
    lib.swa_db_attach.argtypes = [C.c_void_p, C.POINTER(DbView)]
    lib.swa_d1_index_build.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.swa_d1_network.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                   C.c_uint64, u64p]
    lib.swa_d1_network_device.argtypes = lib.swa_d1_network.argtypes
    lib.swa_d1_network_edges_device.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                                C.POINTER(C.c_uint64)]
    lib.swa_d1_set_ownership.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.swa_d1_route_slice.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.swa_d1_index_build_routed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
    lib.swa_d1_route_slice_records.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.swa_d1_index_build_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
    lib.swa_d1_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.swa_d1_table_size.argtypes = [C.c_void_p]
    lib.swa_d1_table_size.restype = C.c_uint64
    lib.swa_d1_fastidious.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.swa_d1_fastidious_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_void_p]
    lib.swa_qgram_build.argtypes = [C.c_void_p]
    lib.swa_qgram_diff.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.swa_qgram_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.swa_search_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.swa_search_do.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
    lib.swa_timing_enable.argtypes = [C.c_void_p, C.c_int]
    lib.swa_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.swa_hostdb_read_fasta.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    lib.swa_hostdb_free.argtypes = [C.c_void_p]
    lib.swa_hostdb_free.restype = None
    lib.swa_hostdb_error.argtypes = [C.c_void_p]
    lib.swa_hostdb_error.restype = C.c_char_p
    lib.swa_hostdb_view.argtypes = [C.c_void_p, C.POINTER(DbView)]
    lib.swa_hostdb_view.restype = None
    lib.swa_hostdb_nucleotides.argtypes = [C.c_void_p]
    lib.swa_hostdb_nucleotides.restype = C.c_uint64
    lib.swa_hostdb_header.argtypes = [C.c_void_p, C.c_uint32, u32p]
    lib.swa_hostdb_header.restype = C.c_char_p
    lib.swa_d1_cluster.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.swa_d1_result_free.argtypes = [C.c_void_p]
    lib.swa_d1_result_free.restype = None
    lib.swa_d1_result_summary.argtypes = [C.c_void_p, u64p]
    lib.swa_d1_result_summary.restype = None
    for fn in (lib.swa_d1_result_swarmid, lib.swa_d1_result_parent, lib.swa_d1_result_generation):
        fn.argtypes = [C.c_void_p]
        fn.restype = C.c_void_p
    lib.swa_d1_light_flags.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, u64p]
    lib.swa_d1_light_flags.restype = C.c_int
    lib.swa_d1_result_detach.argtypes = [C.c_void_p]
    lib.swa_d1_result_detach.restype = C.c_int
    lib.swa_d1_result_error.argtypes = [C.c_void_p]
    lib.swa_d1_result_error.restype = C.c_char_p
    lib.swa_d1_graft.argtypes = [C.c_void_p, C.c_void_p]
    lib.swa_d1_graft.restype = C.c_uint32
    lib.swa_d1_write_swarms.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int64, C.c_int64]
    lib.swa_d1_write_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    lib.swa_d1_write_structure.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    lib.swa_d1_write_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    lib.swa_d1_write_network.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int64]
    _lib = lib
    return lib


def _p64(a: np.ndarray):
    return a.ctypes.data_as(u64p)


def _ptr(a) -> int:
    """Address of a numpy array (host) or of anything with data_ptr() (a torch tensor: device)."""
    if a is None:
        return 0
    if hasattr(a, "data_ptr"):
        return int(a.data_ptr())
    return int(a.ctypes.data)


class HostDb:
    """Packed amplicon database on the host, read from FASTA by the library's own reader
    (mirror of the reference's db_read, src/db.cc:432-803).  Arrays are numpy views into the
    handle's memory (db order)."""

    def __init__(self, path, usearch_abundance: bool = False, append_abundance: int = 0,
                 check_duplicate_sequences: bool = False):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.swa_hostdb_read_fasta(str(path).encode(), int(usearch_abundance), int(append_abundance),
                                            int(check_duplicate_sequences), C.byref(h))
        self.h = h
        if rc != SWA_OK:
            msg = self.lib.swa_hostdb_error(h).decode() if h else "allocation failed"
            self.close()
            raise SwaError(rc, msg)
        u = DbUnorderedView()
        self.lib.swa_hostdb_unordered_view(h, C.byref(u))
        self.n = int(u.n)
        self.longest = int(u.longest)
        self.nucleotides = int(self.lib.swa_hostdb_nucleotides(h))
        self._ordered = None

    @staticmethod
    def _array(ptr, count, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=count)

    def _view(self):
        """The packed sequences contiguous in db order (swa_hostdb_view): gathered by the library on first use."""
        if self._ordered is None:
            v = DbView()
            self.lib.swa_hostdb_view(self.h, C.byref(v))
            seq_off = self._array(v.seq_off, self.n + 1, np.uint64)
            self._ordered = {"seq_off": seq_off, "seqlen": self._array(v.seqlen, self.n, np.uint32),
                             "abundance": self._array(v.abundance, self.n, np.uint64),
                             "seqs": self._array(v.seqs, int(seq_off[self.n]) if self.n else 0, np.uint64)}
        return self._ordered

    seq_off = property(lambda self: self._view()["seq_off"])
    seqlen = property(lambda self: self._view()["seqlen"])
    abundance = property(lambda self: self._view()["abundance"])
    seqs = property(lambda self: self._view()["seqs"])

    def header(self, i: int) -> bytes:
        return self.lib.swa_hostdb_header(self.h, i, None)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.swa_hostdb_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class D1Clusters:
    """Host-side greedy clustering of the d=1 network (mirror of algo_d1_run's host loop,
    src/algod1.cc:1185-1280) with the fastidious bookkeeping and the writers."""

    def __init__(self, hdb: HostDb, offsets: np.ndarray, neighbours: np.ndarray):
        self.lib = load_library()
        self.hdb = hdb
        self._keep = (np.ascontiguousarray(offsets, dtype=np.uint64), np.ascontiguousarray(neighbours, dtype=np.uint32))
        h = C.c_void_p()
        rc = self.lib.swa_d1_cluster(hdb.h, _ptr(self._keep[0]), _ptr(self._keep[1]) if len(self._keep[1]) else 0,
                                     C.byref(h))
        if rc != SWA_OK:
            raise SwaError(rc, "swa_d1_cluster failed")
        self.h = h

    @classmethod
    def from_resident(cls, ctx: "Context", hdb: HostDb, lazy: bool = False, pinned: bool = False) -> "D1Clusters":
        """The same result from the network ctx.d1_network_resident() left in HBM: agglomeration on the GPU
        (swa_d1_cluster_device), per-swarm sums on the host.  lazy = the command line's form (swa_d1_cluster_resident_lazy):
        swarm / generation / parent stay in HBM until asked for; the result then keeps the context alive."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.hdb = hdb
        self._keep = None
        self._ctx = ctx if (lazy or pinned) else None
        if pinned:
            # the command line's two steps: result arrays sized and pinned ahead (swa_d1_result_prepare), then clustered into
            self.lib.swa_d1_result_prepare.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
            self.lib.swa_d1_cluster_resident_prepared.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            h = C.c_void_p()
            rc = self.lib.swa_d1_result_prepare(ctx.h, hdb.h, C.byref(h))
            self.h = h
            if rc == SWA_OK:
                rc = self.lib.swa_d1_cluster_resident_prepared(ctx.h, hdb.h, h)
            if rc != SWA_OK:
                raise SwaError(rc, "swa_d1_cluster_resident_prepared failed: " + ctx.lib.swa_last_error(ctx.h).decode())
            return self
        fn = self.lib.swa_d1_cluster_resident_lazy if lazy else self.lib.swa_d1_cluster_resident
        fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        rc = fn(ctx.h, hdb.h, C.byref(h))
        self.h = h
        if rc != SWA_OK:
            raise SwaError(rc, "swa_d1_cluster_resident failed: " + (self.lib.swa_d1_result_error(h) or ctx.lib.swa_last_error(ctx.h)).decode())
        return self

    def detach(self) -> None:
        """Fetch whatever a lazy result still has on the device; afterwards it does not need its context."""
        rc = self.lib.swa_d1_result_detach(self.h)
        if rc != SWA_OK:
            raise SwaError(rc, "swa_d1_result_detach failed: " + self.lib.swa_d1_result_error(self.h).decode())
        self._ctx = None

    def summary(self) -> dict:
        out = np.zeros(4, dtype=np.uint64)
        self.lib.swa_d1_result_summary(self.h, _p64(out))
        return {"swarms": int(out[0]), "largest": int(out[1]), "maxgen": int(out[2]), "swarms_before_grafting": int(out[3])}

    def _arr(self, fn) -> np.ndarray:
        ptr = fn(self.h)
        n = self.hdb.n
        if not ptr:
            if n == 0:
                return np.zeros(0, dtype=np.uint32)
            raise SwaError(SWA_E_DEVICE, "the clustering's details could not be fetched: " + self.lib.swa_d1_result_error(self.h).decode())
        buf = (C.c_char * (4 * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32, count=n).copy()

    def swarmid(self) -> np.ndarray:
        return self._arr(self.lib.swa_d1_result_swarmid)

    def parent(self) -> np.ndarray:
        return self._arr(self.lib.swa_d1_result_parent)

    def generation(self) -> np.ndarray:
        return self._arr(self.lib.swa_d1_result_generation)

    def light_flags(self, boundary: int = 3):
        flags = np.zeros(self.hdb.n, dtype=np.uint8)
        stats = np.zeros(5, dtype=np.uint64)
        rc = self.lib.swa_d1_light_flags(self.h, boundary, _ptr(flags), _p64(stats))
        if rc != SWA_OK:
            raise SwaError(rc, "swa_d1_light_flags failed: " + self.lib.swa_d1_result_error(self.h).decode())
        return flags, [int(x) for x in stats]

    def graft(self, graft_cand: np.ndarray) -> int:
        g = np.ascontiguousarray(graft_cand, dtype=np.uint32)
        return int(self.lib.swa_d1_graft(self.h, _ptr(g)))

    def write_swarms(self, path, mothur=False, usearch=False, append_abundance=0, differences=1) -> None:
        assert self.lib.swa_d1_write_swarms(self.h, self.hdb.h, str(path).encode(), int(mothur), int(usearch),
                                            append_abundance, differences) == SWA_OK

    def write_stats(self, path, usearch=False) -> None:
        assert self.lib.swa_d1_write_stats(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_structure(self, path, usearch=False) -> None:
        assert self.lib.swa_d1_write_structure(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_seeds(self, path, usearch=False) -> None:
        assert self.lib.swa_d1_write_seeds(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_network(self, path, usearch=False, append_abundance=0) -> None:
        assert self.lib.swa_d1_write_network(self.hdb.h, _ptr(self._keep[0]), _ptr(self._keep[1]),
                                             str(path).encode(), int(usearch), append_abundance) == SWA_OK

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.swa_d1_result_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream.  Mirrors the reference's implicit global state for the path
    (seqindex, hash table, Bloom filters) as an explicit handle."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.swa_ctx_create(device, C.c_void_p(stream or 0), C.byref(h))
        if rc != SWA_OK:
            raise SwaError(rc, "swa_ctx_create failed: no usable gfx950 device (there is no CPU fallback)")
        self.h = h
        self.n = 0
        self._keep = None

    def close(self) -> None:
        if self.h:
            self.lib.swa_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, allow=()) -> int:
        if rc != SWA_OK and rc not in allow:
            raise SwaError(rc, self.lib.swa_last_error(self.h).decode())
        return rc

    def warmup(self, differences: int | None = None) -> None:
        """First-use costs of the device (code objects of the kernel files, copy queues) now, not inside the first call that
        needs them: what the command line does on a helper thread beside the FASTA read (swa_ctx_warmup; with `differences`
        only the kernel files a run at that d launches: swa_ctx_warmup_for)."""
        self.lib.swa_ctx_warmup_for.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.swa_ctx_warmup_for(self.h, -1 if differences is None else int(differences)))

    def synchronize(self) -> None:
        self._check(self.lib.swa_ctx_synchronize(self.h))

    def timing_enable(self, on: bool = True) -> None:
        self._check(self.lib.swa_timing_enable(self.h, int(on)))

    def timing_read_stream(self) -> list:
        """ms of the kernel groups of the streaming d=1 step: keys, partition, groups + lists, pair pass 0, pair pass 1,
        link partition, CSR rows, amplicon lines"""
        ms = (C.c_float * 8)()
        self.lib.swa_timing_read_stream.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self._check(self.lib.swa_timing_read_stream(self.h, ms))
        return [float(x) for x in ms]

    def timing_read(self) -> list:
        ms = (C.c_float * 8)()
        self._check(self.lib.swa_timing_read(self.h, ms))
        return [float(x) for x in ms]

    def upload_hostdb(self, hdb: "HostDb") -> None:
        """The database as the reader keeps it — words in file order — put in db order on the GPU (swa_db_upload_unordered:
        what the command line does)."""
        u = DbUnorderedView()
        self.lib.swa_hostdb_unordered_view(hdb.h, C.byref(u))
        self._check(self.lib.swa_db_upload_unordered(self.h, C.byref(u)))
        self.n = hdb.n

    # ---- L2
    def upload_db(self, seqs, seq_off, seqlen, abundance, longest: int) -> None:
        """Host numpy arrays (db order) -> HBM."""
        n = int(seqlen.shape[0])
        v = DbView(n, int(longest), _ptr(seqs), _ptr(seq_off), _ptr(seqlen), _ptr(abundance))
        self._check(self.lib.swa_db_upload(self.h, C.byref(v)))
        self.n = n

    def attach_db(self, seqs, seq_off, seqlen, abundance, longest: int) -> None:
        """Arrays already in HBM (torch tensors on this GPU); not copied."""
        n = int(seqlen.shape[0])
        v = DbView(n, int(longest), _ptr(seqs), _ptr(seq_off), _ptr(seqlen), _ptr(abundance))
        self._check(self.lib.swa_db_attach(self.h, C.byref(v)))
        self._keep = (seqs, seq_off, seqlen, abundance)
        self.n = n

    # ---- B1
    def d1_index_build(self, first: int = 0, count: int | None = None) -> bool:
        """Returns True when duplicate sequences were found (the reference aborts in that case).
        With a range only the amplicons of [first, first+count) are checked for a twin (multi-GPU:
        every rank its slice, flags combined with a MAX all-reduce)."""
        dup = C.c_int(0)
        if count is None:
            count = self.n - first
        self.lib.swa_d1_index_build_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]
        self._check(self.lib.swa_d1_index_build_range(self.h, first, count, C.byref(dup)), allow=(SWA_E_DUPLICATES,))
        return bool(dup.value)

    def d1_set_ownership(self, rank: int = 0, world: int = 1) -> None:
        """Multi-GPU by ownership: from the next network call on this context finds only its share of
        the links (the anchor groups whose key maps to `rank`, its share of the plain-kernel seeds),
        so a network call returns PARTIAL rows; sharding.exchange_owned_links merges the ranks' links.
        world = 1 restores the complete network."""
        self._check(self.lib.swa_d1_set_ownership(self.h, rank, world))

    def d1_route_slice(self, first: int, count: int, world: int, d_ids, cap: int, d_counts) -> None:
        """Routed multi-GPU index build, step 1 (swa_d1_route_slice): the ids of [first, first+count) grouped by the
        rank that owns their prefix-side / suffix-side key.  d_ids: device int32/uint32 tensor [2 * world * cap],
        d_counts: device tensor [2 * world + 1] (anything with data_ptr()); complete on return."""
        self._check(self.lib.swa_d1_route_slice(self.h, first, count, world, C.c_void_p(d_ids.data_ptr()), cap,
                                                C.c_void_p(d_counts.data_ptr())))

    def d1_index_build_routed(self, d_ids_prefix, n_prefix: int, d_ids_suffix, n_suffix: int) -> bool:
        """Step 3 (swa_d1_index_build_routed): this rank's indexes from the member ids it received; ownership must be
        set.  Returns True when duplicate sequences were found."""
        dup = C.c_int(0)
        self._check(self.lib.swa_d1_index_build_routed(self.h, C.c_void_p(d_ids_prefix.data_ptr() if n_prefix else 0), n_prefix,
                                                       C.c_void_p(d_ids_suffix.data_ptr() if n_suffix else 0), n_suffix, C.byref(dup)),
                    allow=(SWA_E_DUPLICATES,))
        return bool(dup.value)

    def d1_route_slice_records(self, first: int, count: int, world: int, d_records, cap: int, d_counts) -> None:
        """Routed multi-GPU index build with the key records travelling, step 1 (swa_d1_route_slice_records): torch tensors
        d_records int64 [2 * world * cap], d_counts int32 [2 * world + 1]."""
        self._check(self.lib.swa_d1_route_slice_records(self.h, first, count, world, C.c_void_p(d_records.data_ptr()), cap,
                                                        C.c_void_p(d_counts.data_ptr())))

    def d1_index_build_records(self, rec_prefix, rec_suffix) -> bool:
        """Step 3 (swa_d1_index_build_records): this rank's indexes from the key records it received; ownership must be set.
        Returns the duplicate flag of the build (identical sequences inside the rank's prefix groups are met by the
        network call: SwaError SWA_E_DUPLICATES there)."""
        dup = C.c_int(0)
        n_p, n_s = int(rec_prefix.numel()), int(rec_suffix.numel())
        self._check(self.lib.swa_d1_index_build_records(self.h, C.c_void_p(rec_prefix.data_ptr() if n_p else 0), n_p,
                                                        C.c_void_p(rec_suffix.data_ptr() if n_s else 0), n_s, C.byref(dup)),
                    allow=(SWA_E_DUPLICATES,))
        return bool(dup.value)

    def d1_has_duplicates(self, first: int = 0, count: int | None = None) -> bool:
        """The reference's duplicate check over both calls that can meet identical sequences (include/swarm_amd.h): the
        index build when it builds a table, else the network call's prefix pass.  (A test helper: index build + network.)"""
        if self.d1_index_build(first, count):
            return True
        try:
            self.d1_network(False, first, self.n - first if count is None else count)
        except SwaError as e:
            if e.code == SWA_E_DUPLICATES:
                return True
            raise
        return False

    def d1_network(self, no_cluster_breaking: bool = False, first: int = 0, count: int | None = None):
        """CSR over [first, first+count): (offsets u64[count+1], neighbours u32[total]), rows ascending.
        SwaError(SWA_E_DUPLICATES): identical sequences among the seeds' groups."""
        if count is None:
            count = self.n - first
        offsets = np.zeros(count + 1, dtype=np.uint64)
        cap = max(1024, 4 * count)
        total = C.c_uint64(0)
        while True:
            nb = np.zeros(cap, dtype=np.uint32)
            rc = self._check(self.lib.swa_d1_network(self.h, int(no_cluster_breaking), first, count,
                                                     _ptr(offsets), _ptr(nb), cap, C.byref(total)),
                             allow=(SWA_E_CAPACITY,))
            if rc == SWA_OK:
                return offsets, nb[:total.value]
            cap = int(total.value)

    def d1_anchor_windows(self):
        out = np.zeros(2, dtype=np.uint32)
        self.lib.swa_d1_anchor_windows.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.lib.swa_d1_anchor_windows(self.h, _ptr(out)))
        return int(out[0]), int(out[1])

    def d1_anchor_width(self) -> int:
        """Width of the anchor windows the last index build chose, in nucleotides (32, 64 or 128)."""
        self.lib.swa_d1_anchor_width.argtypes = [C.c_void_p]
        self.lib.swa_d1_anchor_width.restype = C.c_uint32
        return int(self.lib.swa_d1_anchor_width(self.h))

    def d1_network_resident(self, no_cluster_breaking: bool = False) -> int:
        """The network of the whole database computed into the context's own HBM buffers and kept there."""
        self.lib.swa_d1_network_resident.argtypes = [C.c_void_p, C.c_int, u64p]
        total = C.c_uint64(0)
        self._check(self.lib.swa_d1_network_resident(self.h, int(no_cluster_breaking), C.byref(total)))
        return int(total.value)

    def d1_network_fetch(self, total: int):
        self.lib.swa_d1_network_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        offsets = np.zeros(self.n + 1, dtype=np.uint64)
        nb = np.zeros(max(total, 1), dtype=np.uint32)
        self._check(self.lib.swa_d1_network_fetch(self.h, _ptr(offsets), _ptr(nb), len(nb)))
        return offsets, nb[:total]

    def d1_network_device(self, d_offsets, d_neighbours, cap: int, no_cluster_breaking: bool = False,
                          first: int = 0, count: int | None = None) -> int:
        """Device-resident CSR (torch tensors); returns the number of neighbours written."""
        if count is None:
            count = self.n - first
        total = C.c_uint64(0)
        self._check(self.lib.swa_d1_network_device(self.h, int(no_cluster_breaking), first, count,
                                                   _ptr(d_offsets), _ptr(d_neighbours), cap, C.byref(total)))
        return int(total.value)

    def d1_guard_retries(self) -> int:
        """Steps this context repeated because the guard's counts did not balance (swa_d1_guard_retries)."""
        self.lib.swa_d1_guard_retries.argtypes = [C.c_void_p]
        return int(self.lib.swa_d1_guard_retries(self.h))

    def d1_network_edges_device(self, d_edge_list, cap: int, no_cluster_breaking: bool = False,
                                first: int = 0, count: int | None = None) -> int:
        """The same links as one flat device list (int64 tensor): source << 32 | target, each once,
        unordered.  Returns the number of links; raises SwaError(SWA_E_CAPACITY) if cap is too small."""
        if count is None:
            count = self.n - first
        total = C.c_uint64(0)
        self._check(self.lib.swa_d1_network_edges_device(self.h, int(no_cluster_breaking), first, count,
                                                         _ptr(d_edge_list), cap, C.byref(total)))
        return int(total.value)

    def d1_table_size(self) -> int:
        return int(self.lib.swa_d1_table_size(self.h))

    def d1_debug(self, what: int, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=np.uint64)
        self._check(self.lib.swa_d1_debug_read(self.h, what, _ptr(out), out.nbytes))
        return out

    # ---- B2
    def d1_fastidious(self, is_light: np.ndarray, light_nt: int, bloom_bits: int = 16, shard: int = 0,
                      nshards: int = 1):
        """B2.  With nshards > 1 only slice `shard` of the heavy amplicons is expanded on this GPU;
        combine the shards with sharding.combine_grafts (minimum of graft_cand, sum of counters 1, 2)."""
        is_light = np.ascontiguousarray(is_light, dtype=np.uint8)
        graft = np.zeros(self.n, dtype=np.uint32)
        counters = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.swa_d1_fastidious_shard(self.h, _ptr(is_light), int(light_nt), int(bloom_bits),
                                                     int(shard), int(nshards), _ptr(graft), _ptr(counters)))
        return graft, counters

    # ---- B3
    def qgram_build(self) -> None:
        self._check(self.lib.swa_qgram_build(self.h))

    def qgram_diff(self, seed: int, amplist: np.ndarray) -> np.ndarray:
        amplist = np.ascontiguousarray(amplist, dtype=np.uint64)
        out = np.zeros(amplist.shape[0], dtype=np.uint64)
        self._check(self.lib.swa_qgram_diff(self.h, int(seed), amplist.shape[0], _ptr(amplist), _ptr(out)))
        return out

    def qgram_signatures(self) -> np.ndarray:
        out = np.zeros((self.n, 128), dtype=np.uint8)
        self._check(self.lib.swa_qgram_debug_read(self.h, _ptr(out), out.nbytes))
        return out

    # ---- B4
    def search_begin(self, mismatch: int = 18, gapopen: int = 24, gapextend: int = 13, d: int = 3) -> None:
        self._check(self.lib.swa_search_begin(self.h, mismatch, gapopen, gapextend, d))

    def search_uses_wavefront(self) -> bool:
        self.lib.swa_search_uses_wavefront.argtypes = [C.c_void_p]
        return bool(self.lib.swa_search_uses_wavefront(self.h))

    def search_do(self, query: int, targets: np.ndarray):
        targets = np.ascontiguousarray(targets, dtype=np.uint64)
        m = targets.shape[0]
        scores = np.zeros(m, dtype=np.uint64)
        diffs = np.zeros(m, dtype=np.uint64)
        alens = np.zeros(m, dtype=np.uint64)
        self._check(self.lib.swa_search_do(self.h, int(query), m, _ptr(targets), _ptr(scores), _ptr(diffs),
                                           _ptr(alens)))
        return scores, diffs, alens


# ---- d >= 2: host greedy loop over the GPU's fused scan step ---------------------------------

_DN_EXPORTS = ["swa_dn_cluster", "swa_dn_result_free", "swa_dn_result_error", "swa_dn_result_summary",
               "swa_dn_write_swarms", "swa_dn_write_stats", "swa_dn_write_structure", "swa_dn_write_seeds",
               "swa_dn_write_uclust", "swa_d1_write_uclust", "swa_scan_begin", "swa_scan_step", "swa_scan_batch", "swa_scan_fetch", "swa_scan_totals",
               "swa_dn_graph_supported", "swa_dn_graph", "swa_dn_graph_totals", "swa_dn_graph_resident", "swa_dn_parent_diffs",
               "swa_multi_create", "swa_multi_destroy", "swa_multi_size", "swa_multi_uses_rccl", "swa_multi_ctx", "swa_multi_last_error",
               "swa_multi_db_upload", "swa_multi_d1_network", "swa_multi_d1_fastidious", "swa_dn_set_ownership", "swa_multi_dn_begin",
               "swa_multi_dn_graph_supported", "swa_multi_dn_graph", "swa_multi_dn_graph_totals", "swa_dn_cluster_multi",
               "swa_timing_read_stream"]
EXPORTS.extend(_DN_EXPORTS)


def reduced_penalties(match_reward: int = 5, mismatch_penalty: int = 4, gap_open: int = 12, gap_extend: int = 4):
    """The reference's scoring reduction (src/swarm.cc:466-483): (2m+2p, 2g, m+2e) / gcd -> 18, 24, 13."""
    from math import gcd
    mm = 2 * match_reward + 2 * mismatch_penalty
    go = 2 * gap_open
    ge = match_reward + 2 * gap_extend
    f = gcd(gcd(mm, go), ge)
    return mm // f, go // f, ge // f


def _declare_dn(lib) -> None:
    if getattr(lib, "_dn_declared", False):
        return
    lib.swa_dn_cluster.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                   C.POINTER(C.c_void_p)]
    lib.swa_dn_result_free.argtypes = [C.c_void_p]
    lib.swa_dn_result_free.restype = None
    lib.swa_dn_result_error.argtypes = [C.c_void_p]
    lib.swa_dn_result_error.restype = C.c_char_p
    lib.swa_dn_result_summary.argtypes = [C.c_void_p, u64p]
    lib.swa_dn_result_summary.restype = None
    lib.swa_dn_write_swarms.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int64]
    for fn in (lib.swa_dn_write_stats, lib.swa_dn_write_structure, lib.swa_dn_write_seeds):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    lib.swa_dn_write_uclust.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int64]
    lib.swa_d1_write_uclust.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint64,
                                        C.c_uint64]
    lib.swa_scan_begin.argtypes = [C.c_void_p]
    lib.swa_scan_step.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_uint32, u32p]
    lib.swa_scan_totals.argtypes = [C.c_void_p, u64p]
    lib._dn_declared = True


class DnClusters:
    """d >= 2 clustering: the reference's algo_run loop (src/algo.cc:384-676) with every
    q-gram / alignment step executed on the GPU (swa_scan_step)."""

    def __init__(self, ctx: "Context", hdb: HostDb, differences: int, no_cluster_breaking: bool = False,
                 penalties=None):
        self.lib = load_library()
        _declare_dn(self.lib)
        self.hdb = hdb
        self.ctx = ctx
        mm, go, ge = penalties or reduced_penalties()
        h = C.c_void_p()
        rc = self.lib.swa_dn_cluster(ctx.h, hdb.h, differences, int(no_cluster_breaking), mm, go, ge, C.byref(h))
        self.h = h
        if rc != SWA_OK:
            raise SwaError(rc, self.lib.swa_dn_result_error(h).decode() if h else "swa_dn_cluster failed")

    def summary(self) -> dict:
        out = np.zeros(3, dtype=np.uint64)
        self.lib.swa_dn_result_summary(self.h, _p64(out))
        return {"swarms": int(out[0]), "largest": int(out[1]), "maxgen": int(out[2])}

    def scan_totals(self) -> dict:
        """Work counters of the route that ran: the bulk graph (swa_dn_graph) or the fused scan (swa_scan_*)."""
        out = np.zeros(3, dtype=np.uint64)
        self.lib.swa_dn_graph_totals.argtypes = [C.c_void_p, u64p]
        self.lib.swa_dn_graph_supported.argtypes = [C.c_void_p]
        if self.lib.swa_dn_graph_supported(self.ctx.h) and os.environ.get("SWARM_AMD_DN") != "scan":
            self.ctx._check(self.lib.swa_dn_graph_totals(self.ctx.h, _p64(out)))
            return {"route": "graph", "qgram_comparisons": int(out[0]), "aligned_pairs": int(out[1]), "launch_sequences": int(out[2])}
        self.ctx._check(self.lib.swa_scan_totals(self.ctx.h, _p64(out)))
        return {"route": "scan", "qgram_comparisons": int(out[0]), "aligned_pairs": int(out[1]), "launch_sequences": int(out[2])}

    def write_swarms(self, path, mothur=False, usearch=False, append_abundance=0) -> None:
        assert self.lib.swa_dn_write_swarms(self.h, self.hdb.h, str(path).encode(), int(mothur), int(usearch),
                                            append_abundance) == SWA_OK

    def write_stats(self, path, usearch=False) -> None:
        assert self.lib.swa_dn_write_stats(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_structure(self, path, usearch=False) -> None:
        assert self.lib.swa_dn_write_structure(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_seeds(self, path, usearch=False) -> None:
        assert self.lib.swa_dn_write_seeds(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_uclust(self, path, usearch=False, append_abundance=0) -> None:
        assert self.lib.swa_dn_write_uclust(self.h, self.hdb.h, str(path).encode(), int(usearch),
                                            append_abundance) == SWA_OK

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.swa_dn_result_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- d = 0: dereplication ----------------------------------------------------------------------

_D0_EXPORTS = ["swa_derep", "swa_d0_cluster", "swa_d0_result_free", "swa_d0_result_summary", "swa_d0_write_swarms",
               "swa_d0_write_seeds", "swa_d0_write_stats", "swa_d0_write_structure", "swa_d0_write_uclust"]
EXPORTS.extend(_D0_EXPORTS)


def _declare_d0(lib) -> None:
    if getattr(lib, "_d0_declared", False):
        return
    lib.swa_derep.argtypes = [C.c_void_p, C.c_void_p]
    lib.swa_d0_cluster.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.swa_d0_result_free.argtypes = [C.c_void_p]
    lib.swa_d0_result_free.restype = None
    lib.swa_d0_result_summary.argtypes = [C.c_void_p, u64p]
    lib.swa_d0_result_summary.restype = None
    lib.swa_d0_write_swarms.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int64, C.c_int64]
    for fn in (lib.swa_d0_write_seeds, lib.swa_d0_write_stats, lib.swa_d0_write_structure):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    lib.swa_d0_write_uclust.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int64]
    lib._d0_declared = True


def derep(ctx: "Context") -> np.ndarray:
    """swa_derep: per amplicon the smallest index with the identical sequence (GPU)."""
    _declare_d0(ctx.lib)
    out = np.zeros(ctx.n, dtype=np.uint32)
    ctx._check(ctx.lib.swa_derep(ctx.h, _ptr(out)))
    return out


class D0Clusters:
    """d = 0 clustering (src/derep.cc) from swa_derep's array; host side only."""

    def __init__(self, hdb: HostDb, first_identical: np.ndarray):
        self.lib = load_library()
        _declare_d0(self.lib)
        self.hdb = hdb
        first_identical = np.ascontiguousarray(first_identical, dtype=np.uint32)
        h = C.c_void_p()
        rc = self.lib.swa_d0_cluster(hdb.h, _ptr(first_identical), C.byref(h))
        if rc != SWA_OK:
            raise SwaError(rc, "swa_d0_cluster failed")
        self.h = h

    def summary(self) -> dict:
        out = np.zeros(3, dtype=np.uint64)
        self.lib.swa_d0_result_summary(self.h, _p64(out))
        return {"swarms": int(out[0]), "largest": int(out[1]), "heaviest": int(out[2])}

    def write_swarms(self, path, mothur=False, usearch=False, append_abundance=0) -> None:
        assert self.lib.swa_d0_write_swarms(self.h, self.hdb.h, str(path).encode(), int(mothur), int(usearch),
                                            append_abundance, 0) == SWA_OK

    def write_seeds(self, path, usearch=False) -> None:
        assert self.lib.swa_d0_write_seeds(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_stats(self, path, usearch=False) -> None:
        assert self.lib.swa_d0_write_stats(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_structure(self, path, usearch=False) -> None:
        assert self.lib.swa_d0_write_structure(self.h, self.hdb.h, str(path).encode(), int(usearch)) == SWA_OK

    def write_uclust(self, path, usearch=False, append_abundance=0) -> None:
        assert self.lib.swa_d0_write_uclust(self.h, self.hdb.h, str(path).encode(), int(usearch),
                                            append_abundance) == SWA_OK

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.swa_d0_result_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def d1_write_uclust(clusters: "D1Clusters", path, usearch=False, append_abundance=0, penalties=None) -> None:
    lib = load_library()
    _declare_dn(lib)
    mm, go, ge = penalties or reduced_penalties()
    assert lib.swa_d1_write_uclust(clusters.h, clusters.hdb.h, str(path).encode(), int(usearch), append_abundance,
                                   mm, go, ge) == SWA_OK


# ---- d = 1 on several GPUs from one process (multi.hip) -----------------------------------------

class MultiContext:
    """swa_multi_*: one context + stream + host thread per listed device inside the library; RCCL for the exchange
    when the devices are distinct, device-to-device copies when a device is listed twice."""

    def __init__(self, devices):
        self.lib = load_library()
        lib = self.lib
        lib.swa_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        lib.swa_multi_destroy.argtypes = [C.c_void_p]
        lib.swa_multi_destroy.restype = None
        lib.swa_multi_last_error.argtypes = [C.c_void_p]
        lib.swa_multi_last_error.restype = C.c_char_p
        lib.swa_multi_uses_rccl.argtypes = [C.c_void_p]
        lib.swa_multi_db_upload.argtypes = [C.c_void_p, C.POINTER(DbView)]
        lib.swa_multi_d1_network.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.POINTER(C.c_int)]
        lib.swa_multi_d1_fastidious.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = lib.swa_multi_create(arr, len(devices), C.byref(h))
        self.h = h
        if rc != SWA_OK:
            msg = lib.swa_multi_last_error(h).decode() if h else "allocation failed"
            self.close()
            raise SwaError(rc, msg)
        self.n = 0

    def _check(self, rc: int, allow=()) -> int:
        if rc != SWA_OK and rc not in allow:
            raise SwaError(rc, self.lib.swa_multi_last_error(self.h).decode())
        return rc

    def uses_rccl(self) -> bool:
        return bool(self.lib.swa_multi_uses_rccl(self.h))

    def upload_hostdb(self, hdb: "HostDb") -> None:
        v = DbView(hdb.n, hdb.longest, _ptr(hdb.seqs), _ptr(hdb.seq_off), _ptr(hdb.seqlen), _ptr(hdb.abundance))
        self._check(self.lib.swa_multi_db_upload(self.h, C.byref(v)))
        self.n = hdb.n

    def d1_network(self, no_cluster_breaking: bool = False):
        offsets = np.zeros(self.n + 1, dtype=np.uint64)
        cap = max(1024, 4 * self.n)
        total = C.c_uint64(0)
        dup = C.c_int(0)
        while True:
            nb = np.zeros(cap, dtype=np.uint32)
            rc = self._check(self.lib.swa_multi_d1_network(self.h, int(no_cluster_breaking), _ptr(offsets), _ptr(nb), cap,
                                                           C.byref(total), C.byref(dup)), allow=(SWA_E_CAPACITY,))
            if rc == SWA_OK:
                return offsets, nb[:total.value]
            cap = int(total.value)

    def dn_graph(self, d: int, no_cluster_breaking: bool = False, mismatch: int = 18, gapopen: int = 24, gapextend: int = 13):
        """swa_multi_dn_begin + swa_multi_dn_graph: (offsets, neighbours, diffs) of the whole d >= 2 graph, or None when a
        sequence is too short for d + 1 windows."""
        lib = self.lib
        lib.swa_multi_dn_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        lib.swa_multi_dn_graph_supported.argtypes = [C.c_void_p]
        lib.swa_multi_dn_graph.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, u64p]
        self._check(lib.swa_multi_dn_begin(self.h, mismatch, gapopen, gapextend, d))
        if not lib.swa_multi_dn_graph_supported(self.h):
            return None
        offsets = np.zeros(self.n + 1, dtype=np.uint64)
        total = C.c_uint64(0)
        rc = self._check(lib.swa_multi_dn_graph(self.h, int(no_cluster_breaking), _ptr(offsets), None, None, 0, C.byref(total)), allow=(SWA_E_CAPACITY,))
        nb = np.zeros(max(1, total.value), dtype=np.uint32)
        df = np.zeros(max(1, total.value), dtype=np.uint8)
        if rc == SWA_E_CAPACITY:
            self._check(lib.swa_multi_dn_graph(self.h, int(no_cluster_breaking), _ptr(offsets), _ptr(nb), _ptr(df), total.value, C.byref(total)))
        return offsets, nb[:total.value], df[:total.value]

    def d1_fastidious(self, is_light: np.ndarray, light_nt: int, bloom_bits: int = 16):
        is_light = np.ascontiguousarray(is_light, dtype=np.uint8)
        graft = np.zeros(self.n, dtype=np.uint32)
        counters = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.swa_multi_d1_fastidious(self.h, _ptr(is_light), int(light_nt), int(bloom_bits), _ptr(graft), _ptr(counters)))
        return graft, counters

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.swa_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
