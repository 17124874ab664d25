#!/usr/bin/env python3
"""bench.py — throughput of the d=1 neighbour-finding path (seam B1) on N MI355X.

One "step" = one full pass of the hot path over the synthetic amplicon set, through the C ABI, with the packed
database already resident in HBM and the CSR left in HBM:
    swa_d1_index_build    amplicon keys -> partition -> anchor groups + work lists (+ identical sequences)
    swa_d1_network_device pair kernels over the groups -> links -> partition by source -> CSR
(swarm_amd/csrc/d1_stream.inc; SWA_D1_BUILD=table / SWA_D1_CSR=table select round 2's hash-table kernels) and, for
N > 1, a MAX all-reduce of the per-rank duplicate flags, the all-to-all merge of the ranks' links by seed range and the
RCCL all-gather of the per-rank CSR slices over xGMI.

Workload at N = 1: the size BASELINE.json's metric string names — 10 M synthetic amplicons x 150 bp, d = 1
(configs[1], 1 M x 150, is measured in the same run and reported under config.configs1).  For N > 1 the job is
weak-scaled: the database holds N x 10 M amplicons (N = 8: 80 M, the order of configs[4]), replicated on every GPU;
rank r serves the anchor groups whose key maps to r (routed index build), the links travel all-to-all by seed range.

At N = 1 the same run also measures, after the timed region (none of it enters `value`; --no-extras skips it):
  config.configs1 / configs2 / configs3   BASELINE configs[1..3]
  config.skewed / skewed_70 / v4_like / heavy_tail   the headline step on conserved flanks (40 nt; 70 nt: wider than a window) /
                                           on 250-nt reads with 60 % conserved positions / on Zipf-sized families
  config.d1_x400                           the d=1 step on 1 M amplicons of 400 bp
  whole_run (top level)                    FASTA -> -o through the drop-in command line, 1 M (md5 vs the reference) and 10 M
  host_seam_ms (top level)                 swa_db_upload + index + swa_d1_network from / to host buffers (PCIe inclusive)
  first_step_ms (top level)                the first index build + network of a fresh context on a resident database (what a real
                                           run pays once: lines, lengths, ranks, the guard's second opinion) beside the repeated step
  configs2 / configs3 (top level)          the rates of config.configs2 / config.configs3 the metric names (q-gram comparisons/s, aligned pairs/s)
  roofline.traffic / roofline.kernels      HBM bytes and VALU instructions per kernel from nested rocprofv3 --pmc passes,
                                           corrected per access pattern as calibrated (profiles/r03/ubench_ceilings_and_pmc_calibration.json)
  roofline.ceilings                        tools/ubench_lines on this box: streaming copy, random lines/s, atomics/s, VALU issue

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline"     — for the dominant kernel (by time) of the step, by the resource that BINDS it.  A pair kernel is bound by
                   VALU issue: bound "valu", achieved = its VALU wave-instructions (SQ_INSTS_VALU of a nested rocprofv3
                   --pmc pass) / its duration (HIP events on the launch stream), peak = the guide's issue rate
                   (256 CUs x 4 SIMDs x 1/2 a cycle x 2.4 GHz = 1.23e12 wave-instr/s), the ceiling tools/ubench_lines measures on
                   this box beside it (peak_measured); roofline.hbm = the HBM view of the same kernel on COUNTER bytes
                   (and on its algorithmic bytes).  Without the counter passes (--no-extras) the object falls back to the
                   HBM view on algorithmic bytes.  The same per kernel group (kernels), the step as a whole (step,
                   step_traffic_over_minimum = measured bytes / (64 n + 12 e + 8 n)); SURVEY.md 8(d)'s figure — the
                   reference's probing loop, which this route never performs — kept as reference_equivalent_rate.
  "cpu_baseline" — the unmodified reference (oracle/_ref/swarm, kind "reference") timed on this box's host cores on a
                   bounded sample (1 M); cpu_baseline_10M: the same on the metric's own 10 M set, once.
"""
from __future__ import annotations

import argparse
import json
import os

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # before torch / the library load libgomp (see swarm_amd/capi.py)


def _usable_cpus() -> int:              # (affinity and cgroup CPU quota: swarm_amd/capi.py usable_cpus, host/pool.h swa_host_cpus)
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, round(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


os.environ.setdefault("OMP_NUM_THREADS", str(min(32, _usable_cpus())))   # (the host phases' OpenMP teams, as the command line sets them)
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue peak of the chip, same guide ("Wave scheduling": 4 SIMD-32 per CU, a 64-lane wave issues each VALU instruction
# over 2 cycles): 256 CUs x 4 SIMDs x 0.5 wave-instructions a cycle x 2.4 GHz
VALU_PEAK_WAVE_INSTR_S = 256 * 4 * 0.5 * 2.4e9


def algorithmic_bytes(seqlen: np.ndarray, hits: int) -> float:
    """per amplicon 8*ceil(L/32) (sequence) + 16 (hash, abundance)
    + V(L)*8 (one 8-byte membership word per variant, V(L) = 6.75 L + 4.25) + 4 per hit."""
    L = seqlen.astype(np.float64)
    words = np.ceil(L / 32.0)
    return float((8.0 * words + 16.0 + (6.75 * L + 4.25) * 8.0).sum() + 4.0 * hits)


def gen_tool() -> Path:
    out = ROOT / "tools" / "gen_amplicons"
    src = ROOT / "tools" / "gen_amplicons.c"
    if not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-o", str(out), str(src), "-lm"], check=True)
    return out


def gen_fasta(n: int, length: int, seed: int, edits: int = 1, light: float = 0.0, flank: int = 0, zipf: float = 0.0, conserved: int = 0) -> Path:
    """The synthetic amplicon set (SURVEY.md section 8d shapes; tools/gen_amplicons.c), cached in
    the temp dir.  Sets above 2 M are generated as independent blocks of <= 2 M amplicons by
    parallel processes (disjoint header numbers, seeds derived from `seed`) and concatenated.
    edits > 1 (the d >= 2 sets) or light > 0 (the --fastidious sets: that fraction of abundance-1
    amplicons 2-3 edits away from a heavy one) are one generator call whatever the size."""
    if edits != 1 or light != 0.0:
        fasta = Path(tempfile.gettempdir()) / f"swa_cfg_{n}x{length}_s{seed}_e{edits}_l{light}.fa"
        if not fasta.exists():
            tmp = fasta.with_suffix(f".tmp{os.getpid()}")
            subprocess.run([str(gen_tool()), str(n), str(length), str(seed), str(edits), str(light), str(tmp)], check=True)
            os.replace(tmp, fasta)
        return fasta
    # flank > 0: all centroids share their first and last `flank` nucleotides (GEN_FLANK of the generator): the skewed
    # case for anchors at the ends of the sequences
    # zipf > 0: family sizes follow Zipf's law (GEN_ZIPF): the largest family of every 2 M block is that share of the block
    # conserved > 0: that per cent of the positions, in stretches of 8..40 nt, is the same in every centroid (GEN_CONSERVED)
    tag = (f"_f{flank}" if flank else "") + (f"_z{zipf}" if zipf else "") + (f"_c{conserved}" if conserved else "")
    fasta = Path(tempfile.gettempdir()) / f"swa_bench_{n}x{length}_s{seed}{tag}.fa"
    if fasta.exists():
        return fasta
    genv = dict(os.environ, **({"GEN_FLANK": str(flank)} if flank else {}), **({"GEN_ZIPF": str(zipf)} if zipf else {}),
                **({"GEN_CONSERVED": str(conserved)} if conserved else {})) if (flank or zipf or conserved) else None
    tool = str(gen_tool())
    tmp = fasta.with_suffix(f".tmp{os.getpid()}")
    block = 2_000_000
    if n <= block:
        subprocess.run([tool, str(n), str(length), str(seed), "1", "0", str(tmp)], check=True, env=genv)
    else:
        parts, procs, at, b = [], [], 0, 0
        limit = max(1, min(32, (os.cpu_count() or 2) // 2))
        while at < n:
            size = min(block, n - at)
            part = fasta.with_suffix(f".part{b}.{os.getpid()}")
            parts.append(part)
            procs.append(subprocess.Popen([tool, str(size), str(length), str(seed * 1000 + b + 1), "1", "0", str(part), str(at)], env=genv))
            at += size
            b += 1
            while sum(p.poll() is None for p in procs) >= limit:
                time.sleep(0.05)
        for p in procs:
            if p.wait() != 0:
                raise SystemExit("gen_amplicons failed")
        with open(tmp, "wb") as out:
            for part in parts:
                with open(part, "rb") as src:
                    while True:
                        chunk = src.read(64 << 20)
                        if not chunk:
                            break
                        out.write(chunk)
                part.unlink()
    os.replace(tmp, fasta)
    return fasta


def cpu_baseline(fasta: Path, n: int, length: int, seed: int) -> dict:
    """Reference swarm (or, failing that, the C oracle) on this box's host cores, bounded sample."""
    cores = os.cpu_count() or 1
    ref = ROOT / "oracle" / "_ref" / "swarm"
    sample_n = n if cores >= 8 else max(100_000, n // 4)
    with tempfile.TemporaryDirectory() as tmp:
        sample = fasta
        if sample_n != n:
            sample = Path(tmp) / "sample.fa"
            subprocess.run([str(gen_tool()), str(sample_n), str(length), str(seed), "1", "0", str(sample)], check=True)
        if ref.exists():
            # the reference scales to about 8-16 threads at 150 bp (its README): try a few thread
            # counts within the 10-30 s budget and report the best one
            best = None
            tried = []
            for threads in sorted({min(cores, t) for t in (8, 16, 32)}):
                t0 = time.perf_counter()
                subprocess.run([str(ref), "-d", "1", "-t", str(threads), "-o", f"{tmp}/ref.o", "-l", "/dev/null",
                                str(sample)], check=True)
                dt = time.perf_counter() - t0
                tried.append(f"-t {threads}: {dt:.2f} s")
                if best is None or dt < best[1]:
                    best = (threads, dt)
            threads, dt = best
            return {"output_md5": md5_of(f"{tmp}/ref.o"), "value": sample_n / dt, "unit": "amplicons/s", "cores": threads, "kind": "reference",
                    "seconds": round(dt, 2), "tried": "; ".join(tried), "host_cores": cores,
                    "sample": f"reference swarm 3.1.6 -d 1 -t {threads}, whole run (FASTA to -o) on {sample_n} x {length} bp, best of -t 8/16/32"}
        # port: the single-threaded C oracle's network construction (test infrastructure, timed only)
        sys.path.insert(0, str(ROOT / "tests"))
        import support as S
        from swarm_amd import HostDb
        hdb = HostDb(sample)
        db = S.Db(headers=[], seqs=hdb.seqs, seq_off=hdb.seq_off, seqlen=hdb.seqlen, abundance=hdb.abundance,
                  longest=hdb.longest)
        t0 = time.perf_counter()
        S.oracle_d1_network(db)
        dt = time.perf_counter() - t0
        return {"value": sample_n / dt, "unit": "amplicons/s", "cores": 1, "kind": "port",
                "sample": f"oracle/ C restatement, index build + network only, {sample_n} x {length} bp, {dt:.2f} s"}


def gen_mixed_fasta(n: int, length: int, seed: int) -> Path:
    """n x length amplicons plus one per cent of 420-480 nt ones (five blocks of lengths 425, 438, 450, 462, 475, each with
    its own families): the file a 16S V4 run with a few V3-V4 or chimeric reads in it looks like."""
    fasta = Path(tempfile.gettempdir()) / f"swa_bench_{n}x{length}_s{seed}_mixed.fa"
    if fasta.exists():
        return fasta
    base = gen_fasta(n, length, seed)
    tmp = fasta.with_suffix(f".tmp{os.getpid()}")
    tool = str(gen_tool())
    with open(tmp, "wb") as out:
        with open(base, "rb") as src:
            while True:
                chunk = src.read(64 << 20)
                if not chunk:
                    break
                out.write(chunk)
        per = max(1, n // 500)
        for b, lng in enumerate((425, 438, 450, 462, 475)):
            part = fasta.with_suffix(f".long{b}.{os.getpid()}")
            subprocess.run([tool, str(per), str(lng), str(seed * 7919 + b), "1", "0", str(part), str(n + b * per)], check=True)
            out.write(part.read_bytes())
            part.unlink()
    os.replace(tmp, fasta)
    return fasta


def extra_measurement(torch, dev, device_index: int, args, n: int, steps: int, flank: int = 0, zipf: float = 0.0, mixed: bool = False, conserved: int = 0) -> dict:
    """The bench step (index build + network, db and CSR resident) on n x length amplicons."""
    from swarm_amd import Context, HostDb
    hdb = HostDb(gen_mixed_fasta(n, args.length, args.seed) if mixed else gen_fasta(n, args.length, args.seed, 1, 0.0, flank, zipf, conserved))

    def to_dev(a: np.ndarray, as_dtype):
        return torch.from_numpy(np.ascontiguousarray(a).view(as_dtype)).to(dev)

    t_seqs = to_dev(np.concatenate([hdb.seqs, np.zeros(2, dtype=np.uint64)]), np.int64)
    t_off, t_len, t_ab = to_dev(hdb.seq_off, np.int64), to_dev(hdb.seqlen, np.int32), to_dev(hdb.abundance, np.int64)
    ctx = Context(device_index, torch.cuda.current_stream(dev).cuda_stream)
    ctx.attach_db(t_seqs, t_off, t_len, t_ab, hdb.longest)
    ctx.timing_enable(True)
    cap = 8 * hdb.n
    d_offsets = torch.zeros(hdb.n + 1, dtype=torch.int64, device=dev)
    d_nb = torch.zeros(cap, dtype=torch.int32, device=dev)
    total, k_ms = 0, []
    warm = 2                                           # (the second step still sizes buffers the first step's counts ask for)
    for it in range(warm + steps):
        if it == warm:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        assert not ctx.d1_index_build()
        total = ctx.d1_network_device(d_offsets, d_nb, cap, False, 0, hdb.n)
        if it >= warm:
            k_ms.append(ctx.timing_read()[3])
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    k = float(np.mean(k_ms))
    abytes = algorithmic_bytes(hdb.seqlen, total)
    windows = ctx.d1_anchor_windows()
    width = ctx.d1_anchor_width()
    t8, st = ctx.timing_read(), ctx.timing_read_stream()
    ctx.close()
    nucleotides = float(hdb.seqlen.astype(np.int64).sum())
    return {"workload": f"{hdb.n} synthetic amplicons x {args.length} bp, d=1" + (f", all centroids share their first / last {flank} nt" if flank else "")
            + (", of them one per cent 420-480 nt long" if mixed else "")
            + (f", {conserved} % of the positions (stretches of 8-40 nt) the same in every centroid" if conserved else "")
            + (f", Zipf family sizes (GEN_ZIPF={zipf}: the largest family of every 2 M block holds that share of it)" if zipf else ""),
            "anchor_windows_nt_from_the_ends": list(windows), "anchor_width_nt": width, "value": hdb.n * steps / elapsed,
            "ns_per_nucleotide": 1e9 * (elapsed / steps) / nucleotides,
            "unit": "amplicons/s", "steps": steps, "ms_per_step": 1000.0 * elapsed / steps, "network_kernels_ms": k,
            "neighbour_links": int(total),
            "kernel_group_ms": {"keys": st[0], "partition_keys": st[1], "groups": st[2], "pairs0": st[3], "pairs1": st[4], "partition_links": st[5], "csr_rows": st[6],
                                # (members of groups beyond the pair kernels' limit: their hash table + Bloom filter at index build,
                                # and the enumerating plain kernel inside the network call — what is left of it beside the pair passes)
                                "plain_kernel_and_table": t8[0] + t8[1], "plain_kernel_in_network": max(0.0, k - st[3] - st[4])},
            # (SURVEY 8(d) bytes — the reference's probing loop — over the step's wall time: an equivalent rate, see roofline)
            "reference_equivalent_GBs": abytes / (elapsed / steps) / 1e9}


def first_step(torch, dev, device_index: int, args, n: int) -> dict:
    """What a real run pays ONCE: the first index build + network on a database that has just
    arrived in HBM — k_lines_build, the lengths / anchor sample / abundance ranks, the guard's second opinion
    (k_guard_db, k_guard_records) and the step itself — next to the same step repeated.  Code objects are loaded (this
    process has run the step before), buffers are NOT: a fresh context allocates everything inside the timed call."""
    from swarm_amd import Context, HostDb
    hdb = HostDb(gen_fasta(n, args.length, args.seed))

    def to_dev(a: np.ndarray, as_dtype):
        return torch.from_numpy(np.ascontiguousarray(a).view(as_dtype)).to(dev)

    t_seqs = to_dev(np.concatenate([hdb.seqs, np.zeros(2, dtype=np.uint64)]), np.int64)
    t_off, t_len, t_ab = to_dev(hdb.seq_off, np.int64), to_dev(hdb.seqlen, np.int32), to_dev(hdb.abundance, np.int64)
    cap = 8 * hdb.n
    d_offsets = torch.zeros(hdb.n + 1, dtype=torch.int64, device=dev)
    d_nb = torch.zeros(cap, dtype=torch.int32, device=dev)
    firsts, repeats = [], []
    for _ in range(3):
        ctx = Context(device_index, torch.cuda.current_stream(dev).cuda_stream)
        ctx.attach_db(t_seqs, t_off, t_len, t_ab, hdb.longest)
        ms = []
        for it in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            assert not ctx.d1_index_build()
            ctx.d1_network_device(d_offsets, d_nb, cap, False, 0, hdb.n)
            torch.cuda.synchronize(dev)
            ms.append(1e3 * (time.perf_counter() - t0))
        firsts.append(ms[0])
        repeats.append(min(ms[1:]))
        ctx.close()
    return {"first_step_ms": round(min(firsts), 3), "repeated_step_ms": round(min(repeats), 3), "first_step_ms_all": [round(x, 3) for x in firsts],
            "what": f"{hdb.n} x {args.length} bp resident in HBM, fresh context: first swa_d1_index_build + swa_d1_network_device "
                    "(buffer allocation, k_lines_build, lengths, anchor sample, abundance ranks, guard second opinion, one step) "
                    "against the same two calls repeated; best of 3 contexts, wall clock around synchronised calls"}


def md5_of(path) -> str:
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as fh:
        while True:
            chunk = fh.read(64 << 20)
            if not chunk:
                break
            h.update(chunk)
    return h.hexdigest()


def reference_run(flags: list, fasta: Path, n: int, what: str) -> dict:
    """The unmodified reference (oracle/_ref/swarm) on this box's host cores, whole run, once: the stated baseline of a
    config other than the headline's.  -t 16: what the headline's baseline found best on this host."""
    ref = ROOT / "oracle" / "_ref" / "swarm"
    if not ref.exists():
        return {"error": "oracle/_ref/swarm not built (no /root/reference at build time)"}
    threads = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    subprocess.run([str(ref), *flags, "-t", str(threads), "-o", "/dev/null", "-l", "/dev/null", str(fasta)], check=True)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "amplicons/s", "cores": threads, "kind": "reference", "seconds": round(dt, 2),
            "sample": f"reference swarm 3.1.6 {' '.join(flags)} -t {threads}, whole run, {what}"}


def config2_fastidious(args, n: int) -> dict:
    """BASELINE configs[2]: n x 150 with 30 % light amplicons, d=1 --fastidious, whole pipeline through the
    C ABI as the command line drives it (network kept in HBM, agglomeration on the GPU, fastidious pair route);
    the two fastidious kernel groups timed with HIP events.  SURVEY 8(d) bytes of the
    fastidious pass: 8 B per light microvariant (written) + 8 B per heavy microvariant (read); the second
    level (p V V' 8) is left out, so the fraction is a lower bound."""
    from swarm_amd import Context, D1Clusters, HostDb
    T = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        r = fn()
        T[name] = round(time.perf_counter() - t0, 4)
        return r

    fa = gen_fasta(n, args.length, args.seed, 1, 0.3)
    hdb = timed("fasta_read_sort_pack", lambda: HostDb(fa))
    ctx = Context(0)
    ctx.timing_enable(True)
    timed("upload", lambda: ctx.upload_hostdb(hdb))
    timed("index_build", ctx.d1_index_build)
    timed("network_resident", ctx.d1_network_resident)
    cl = timed("clustering_gpu_plus_sums", lambda: D1Clusters.from_resident(ctx, hdb))
    flags, stats = timed("light_flags", cl.light_flags)
    graft, counters = timed("fastidious_gpu", lambda: ctx.d1_fastidious(flags, stats[2]))
    ms = ctx.timing_read()
    grafts = timed("graft", lambda: cl.graft(graft))
    out = Path(tempfile.gettempdir()) / f"swa_bench_cfg2_{os.getpid()}.out"
    timed("write_swarms", lambda: cl.write_swarms(out))
    out.unlink()
    k_ms = float(ms[5] + ms[6])
    # Own bytes of the fastidious kernels (the SURVEY 8(d) figure — 8 B per light / heavy microvariant — prices
    # the reference's Bloom probing, which the pair route never does; it survives below as reference_equivalent_rate):
    # every amplicon's line once per group kind (three group builds: prefix, suffix, middle windows) + its role byte and
    # group records (64 + 1 + 12 bytes), and per candidate pair that reaches the intersection count two lines and the
    # 8-byte pair record.  The pass is instruction- and latency-bound, not a memory pass: the fraction says so.
    own = 3.0 * hdb.n * (64.0 + 1.0 + 12.0) + float(counters[2]) * (2 * 64.0 + 8.0)
    abytes = 8.0 * (float(counters[0]) + float(counters[1]))
    res = {"workload": f"{hdb.n} synthetic amplicons x {args.length} bp (30 % light), d=1 --fastidious",
           "pipeline_seconds": T, "pipeline_total_s": round(sum(T.values()), 3),
           "value": hdb.n / sum(T.values()), "unit": "amplicons/s (FASTA -> swarms file, whole pipeline)",
           "fastidious_kernels_ms": {"groups_and_pairs": float(ms[5]), "count_intersections": float(ms[6])},
           "light_variants": int(counters[0]), "heavy_variants": int(counters[1]), "graft_candidates": int(counters[2]),
           "grafts": int(grafts), "swarms": cl.summary()["swarms"],
           "roofline": {"bound": "hbm", "kernel": "the fastidious kernels (three group builds, k_fast_pairs_lines, k_fast_count_sites)",
                        "algorithmic_bytes": own, "achieved": own / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": own / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
                        "note": "integer-VALU / latency bound (profiles/r03/config2_10M_fastidious_kernels_pmc.json: k_fast_pairs_lines at "
                                "3.1-3.8e11 wave-instructions/s, 0.25-0.31 of the 1.23e12 issue peak); own bytes, not SURVEY 8(d)'s",
                        "reference_equivalent_rate": {"bytes": abytes, "GB/s": abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                                                      "note": "SURVEY 8(d): 8 B per light + per heavy microvariant — the reference's Bloom probing, "
                                                              "which this route does not perform; an equivalent rate, not a roofline fraction"}}}
    gold = ROOT / "tests" / "golden" / "fullsize.json"
    if gold.exists():                                   # the reference's own numbers for this set (tests/golden/make_fullsize.py)
        g = json.loads(gold.read_text()).get(str(n), {}).get("runs", {}).get("d1_f")
        if g:
            log = " ".join(g["log"])
            res["counters_equal_reference_log"] = all(str(v) in log for v in (int(counters[0]), int(counters[1]), int(counters[2]), int(grafts)))
    cl.close()
    ctx.close()
    hdb.close()
    if not args.no_cpu_baseline:
        # bounded sample (13 s at -t 8 for 1 M): the same generator, 1 M amplicons with 30 % light ones
        sample = min(n, 1_000_000)
        res["cpu_baseline"] = reference_run(["-d", "1", "-f"], gen_fasta(sample, args.length, args.seed, 1, 0.3), sample,
                                            f"{sample} x {args.length} bp, 30 % light")
    return res


def config3_dn(args, n: int, length: int, d: int) -> dict:
    """BASELINE configs[3]: n x 400 bp, d=3 — q-gram prefilter + alignment scan (B3 + B4) driving the host's greedy
    loop.  Rates over the whole clustering phase: q-gram comparisons/s (144 B each: 128 B signature +
    8 B id + 8 B out), aligned pairs/s, DP cells/s as the reference's full matrices and as the band an accepted
    pair can need."""
    from swarm_amd import Context, DnClusters, HostDb
    fa = gen_fasta(n, length, args.seed, d, 0.0)
    t0 = time.perf_counter()
    hdb = HostDb(fa, check_duplicate_sequences=True)
    t_read = time.perf_counter() - t0
    ctx = Context(0)
    ctx.warmup()                                  # (code objects loaded beside the FASTA read, as the command line does it)
    ctx.timing_enable(True)
    ctx.upload_hostdb(hdb)
    t0 = time.perf_counter()
    cl = DnClusters(ctx, hdb, d)
    dt = time.perf_counter() - t0
    ms = ctx.timing_read()
    scan = cl.scan_totals()
    mm, go, ge = 18, 24, 13
    band = 2 * ((d * max(mm, go + ge)) // ge + 1) + 1
    res = {"workload": f"{hdb.n} synthetic amplicons x {length} bp, d={d}", "route": scan["route"],
           "clustering_seconds": round(dt, 3), "fasta_read_seconds": round(t_read, 3), "value": hdb.n / dt,
           "unit": "amplicons/s (clustering phase: graph of pairs within d and the greedy walk on the GPU, swarm tables on the host)",
           "swarms": cl.summary()["swarms"], "kernel_launches": scan["launch_sequences"],
           "qgram_comparisons": scan["qgram_comparisons"], "aligned_pairs": scan["aligned_pairs"],
           "qgram_comparisons_per_s": scan["qgram_comparisons"] / dt, "aligned_pairs_per_s": scan["aligned_pairs"] / dt,
           "full_matrix_equivalent_cells_per_s": scan["aligned_pairs"] * float(length) * length / dt,
           "banded_cells_per_s": scan["aligned_pairs"] * float(band) * length / dt}
    if scan["route"] == "graph" and ms[6] > 0:
        # the two kernel groups of dn_graph.hip by HIP events; the alignment group is integer VALU / LDS work
        # (reported as pairs and cells per second of kernel time), the pair group streams signatures
        a_s, p_s = ms[6] * 1e-3, ms[5] * 1e-3
        pbytes = 272.0 * scan["qgram_comparisons"]          # two 128-byte signatures + two ids per comparison
        res["gpu_kernels_ms"] = {"groups_and_pairs": float(ms[5]), "alignments_and_csr": float(ms[6])}
        res["alignment_kernel_pairs_per_s"] = scan["aligned_pairs"] / a_s
        res["alignment_kernel_full_matrix_equivalent_cells_per_s"] = scan["aligned_pairs"] * float(length) * length / a_s
        # k_align_wfa by its own bytes (both sequences of a pair, 8-byte result) and by full-matrix-equivalent cells; the
        # kernel is integer-VALU / LDS work (profiles/r03/config3_1M_x400_d3_kernels_pmc.json: 1.8e11 wave-instructions/s)
        wbytes = scan["aligned_pairs"] * (2.0 * 8.0 * ((length + 31) // 32) + 8.0)
        res["roofline"] = {"bound": "hbm", "kernel": "k_dg_pairs (group bookkeeping + q-gram signatures of the pairs)",
                           "algorithmic_bytes": pbytes, "achieved": pbytes / p_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": pbytes / p_s / 1e9 / HBM_PEAK_GBS,
                           "kernels": {"groups_and_pairs": {"ms": float(ms[5]), "algorithmic_bytes": pbytes, "GB/s": pbytes / p_s / 1e9,
                                                            "frac_of_hbm_peak": pbytes / p_s / 1e9 / HBM_PEAK_GBS},
                                       "alignments_and_csr": {"ms": float(ms[6]), "algorithmic_bytes": wbytes, "GB/s": wbytes / a_s / 1e9,
                                                              "frac_of_hbm_peak": wbytes / a_s / 1e9 / HBM_PEAK_GBS,
                                                              "full_matrix_equivalent_cells_per_s": scan["aligned_pairs"] * float(length) * length / a_s,
                                                              "note": "integer VALU / LDS bound, not a memory kernel: cells/s is the rate that means something"}}}
    else:
        qbytes = 144.0 * scan["qgram_comparisons"]
        res["roofline"] = {"bound": "hbm", "kernel": "q-gram scan over the clustering phase (launch-latency bound)",
                           "algorithmic_bytes": qbytes, "achieved": qbytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": qbytes / dt / 1e9 / HBM_PEAK_GBS}
    cl.close()
    ctx.close()
    hdb.close()
    if not args.no_cpu_baseline:
        # The reference's d >= 2 loop is quadratic in the size of the set (20 k x 400 takes 5.7 s, 1 M x 400
        # 1 061 s at -t 8, tests/golden/fullsize.json): a 50 k sample of the same generator keeps it inside the budget, and
        # FLATTERS the reference — its rate at 1 M is a twentieth of the sample's.
        sample = min(n, 50_000)
        res["cpu_baseline"] = reference_run(["-d", str(d)], gen_fasta(sample, length, args.seed, d, 0.0), sample,
                                            f"{sample} x {length} bp (quadratic: slower per amplicon at {n})")
    return res


def host_seam(args, n: int) -> dict:
    """The B1 seam as INTEGRATION.md binds it: host buffers in (swa_db_upload), CSR out on the host (swa_d1_network)."""
    from swarm_amd import Context, HostDb
    hdb = HostDb(gen_fasta(n, args.length, args.seed))
    ctx = Context(0)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.upload_hostdb(hdb)
        t1 = time.perf_counter()
        assert not ctx.d1_index_build()
        t2 = time.perf_counter()
        off, nb = ctx.d1_network()
        t3 = time.perf_counter()
        cur = {"upload_ms": 1e3 * (t1 - t0), "index_build_ms": 1e3 * (t2 - t1), "network_incl_download_ms": 1e3 * (t3 - t2),
               "total_ms": 1e3 * (t3 - t0)}
        if best is None or cur["total_ms"] < best["total_ms"]:
            best = cur
    best["amplicons_per_s"] = hdb.n / (best["total_ms"] * 1e-3)
    best["workload"] = f"{hdb.n} x {args.length} bp, d=1, host buffers in, CSR on the host out (pageable memory), best of 3"
    ctx.close()
    hdb.close()
    return best


def whole_run(args, n: int, ref_out_md5: str | None, sample_n: int) -> dict:
    """FASTA -> -o through the drop-in command line (swarm_amd/bin/swarm): the number BASELINE.md compares
    whole-run vs whole-run; on the cpu_baseline's sample the output must be the reference's, byte for byte."""
    exe = ROOT / "swarm_amd" / "bin" / "swarm"
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for size in (sample_n, n):
            fa = gen_fasta(size, args.length, args.seed)
            # the metric's own size: a distribution, not a best case (the run is bimodal — in some runs the
            # exiting process takes its address space apart itself); the sample: three runs for the md5
            count = args.whole_runs if size == n and n > sample_n else 3
            runs = []
            for _ in range(count):
                time.sleep(0.7)                              # (the previous process's GPU-side teardown is the kernel's, and asynchronous)
                t0 = time.perf_counter()
                subprocess.run([str(exe), "-d", "1", "-o", f"{tmp}/o", "-l", "/dev/null", str(fa)], check=True)
                runs.append(time.perf_counter() - t0)
            ordered = sorted(runs)
            median = ordered[len(ordered) // 2]
            res[f"n{size}"] = {"seconds": round(median, 3), "median_s": round(median, 3), "best_s": round(ordered[0], 3),
                               "p95_s": round(ordered[min(len(ordered) - 1, int(np.ceil(0.95 * len(ordered))) - 1)], 3), "max_s": round(ordered[-1], 3),
                               "runs": len(ordered), "amplicons_per_s": size / median, "all_runs_seconds": [round(x, 3) for x in runs]}
            if size == sample_n and ref_out_md5 is not None:
                res[f"n{size}"]["output_md5_equals_reference"] = md5_of(f"{tmp}/o") == ref_out_md5
    res["what"] = ("swarm_amd/bin/swarm -d 1 -o, one process: process start, FASTA read + sort + pack, upload, index, network, "
                   "agglomeration on the GPU, write, process exit; `seconds` = the MEDIAN of `runs` runs (p95_s, max_s beside it)")
    return res


KERNEL_GROUPS = [          # (substring of the kernel name, group, FETCH_SIZE correction: 2 streaming / 1 random lines)
    ("k_keys", "keys", 2.0), ("k_lines_build", "lines", 2.0),
    ("k_part_", "partition", 2.0), ("k_flat_", "partition", 2.0), ("k_seg_starts", "partition", 2.0),      # (both partitions: the counters
    # cannot tell the key records' launches from the links')
    ("k_group_lists", "lists", 2.0), ("k_group", "groups", 2.0), ("k_clear_many", "clears", 2.0),
    ("k_d1_group_pairs<0", "pairs0", 1.0), ("k_d1_group_pairs<1", "pairs1", 1.0), ("k_d1_pairs_tiled<0", "pairs0", 1.0),
    ("k_d1_pairs_tiled<1", "pairs1", 1.0), ("k_csr_bucket", "csr_rows", 2.0), ("k_seg_reduce", "csr_rows", 2.0),
    # round 2's kernels (SWA_D1_BUILD=table / SWA_D1_CSR=table)
    ("k_anchor_place", "table_index", 1.0), ("k_anchor_scatter", "table_index", 1.0), ("k_scan_", "table_index", 2.0),
    ("k_anchor_clear", "table_index", 2.0), ("k_dup_", "table_index", 1.0), ("k_scatter_edges", "table_csr", 1.0),
    ("k_sort_", "table_csr", 2.0),
]


def kernel_group(name: str):
    for sub, group, corr in KERNEL_GROUPS:
        if sub in name:
            return group, corr
    return ("other", 2.0) if "anonymous namespace" in name else (None, 2.0)


def measured_pmc(args) -> dict | None:
    """Per kernel group of one bench step, from nested rocprofv3 passes over a short run of this very script (separate
    --pmc runs: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU): HBM bytes and VALU wave-instructions.  FETCH_SIZE is reported
    in 64-byte requests: a wide streaming read shows half its bytes (x 2, MI355X_MICROARCH.md), a random line read
    shows 64 bytes per access (x 1) — calibrated with tools/ubench_lines (profiles/r03); the pair kernels' fetches are
    random lines, everything else streams.  WRITE_SIZE is taken as reported (exact for streaming stores)."""
    import csv
    import glob
    import shutil
    if shutil.which("rocprofv3") is None:
        return None
    steps, warm = 2, 1
    groups: dict = {}
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", f"{tmp}/{counter}", "-o", "t", "--",
                   sys.executable, str(ROOT / "bench.py"), "--steps", str(steps), "--warmup", str(warm), "--no-extras",
                   "--per-gpu", str(args.per_gpu), "--length", str(args.length), "--seed", str(args.seed)]
            try:
                subprocess.run(cmd, check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            except Exception:
                return None
            rows = 0
            for f in glob.glob(f"{tmp}/{counter}/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] != counter:
                            continue
                        group, corr = kernel_group(row["Kernel_Name"])
                        if group is None or group == "lines":          # (the lines are made once per upload, not per step)
                            continue
                        g = groups.setdefault(group, {"fetch_bytes": 0.0, "write_bytes": 0.0, "valu_wave_instructions": 0.0, "launches": 0.0})
                        if counter == "SQ_INSTS_VALU":                  # (one row per dispatch in a single-counter pass)
                            g["launches"] += 1.0 / (steps + warm)
                        v = float(row["Counter_Value"]) / (steps + warm)
                        if counter == "FETCH_SIZE":
                            g["fetch_bytes"] += v * 1024.0 * corr
                        elif counter == "WRITE_SIZE":
                            g["write_bytes"] += v * 1024.0
                        else:
                            g["valu_wave_instructions"] += v
                        rows += 1
            if rows == 0:
                return None
    for g in groups.values():
        g["hbm_bytes"] = g["fetch_bytes"] + g["write_bytes"]
    step_groups = ("keys", "partition", "groups", "lists", "pairs0", "pairs1", "csr_rows", "clears")
    return {"per_kernel_group": groups, "hbm_bytes_per_step": sum(g["hbm_bytes"] for g in groups.values()),
            # (launches per step — the library's own kernels; the runtime's fills and copies are not counted)
            "kernel_launches_per_step": round(sum(g.get("launches", 0.0) for k, g in groups.items() if k in step_groups), 1),
            "how": f"nested rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) over bench.py --steps {steps} --warmup {warm}, per step; "
                   "FETCH_SIZE x 2 for the streaming kernels, x 1 for the pair kernels (random 64-byte lines), as calibrated by tools/ubench_lines"}


def measured_ceilings() -> dict | None:
    """tools/ubench_lines (quick): what this box's memory system and VALUs deliver for the access patterns of the step."""
    exe = ROOT / "tools" / "ubench_lines"
    if not exe.exists():
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.run([str(exe), f"{tmp}/u.json", "quick"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            res = json.loads(Path(f"{tmp}/u.json").read_text())["results"]
    except Exception:
        return None
    out = {}
    for r in res:
        out[r["test"]] = {"rate": r["rate"], "unit": r["unit"].split(" ")[0], "working_set_mb": r["working_set_mb"]}
    return out


def step_byte_model(n: int, links: int, index_levels: int, link_levels: int) -> dict:
    """Algorithmic bytes of the kernel groups of one streaming step: what each group has to read
    and write given its inputs and outputs — 64-byte amplicon lines, 8-byte key records (round 6: no fingerprints travel
    with the prefix index any more), 4-byte member ids, 8-byte links."""
    n, e = float(n), float(links)
    return {
        "keys": 64 * n + 16 * n,                                   # lines in; two record arrays out
        # per level: scatter in / out (two record sets) + a histogram pass over the records — which the first level
        # does not have: k_keys takes its histogram on the way
        "partition_keys": index_levels * (2 * 16 * n) + (index_levels - (0 if os.environ.get("SWA_D1_KEYS_HIST", "")[:1] == "0" else 1)) * 16 * n,
        "partition_links": link_levels * (8 * e + 2 * 8 * e),
        "groups": 16 * n + 8 * n,                                  # records in, members out (work items: a few %)
        "pairs0": 68 * n + 4 * e,                                  # id + line per member; half the links out
        "pairs1": 68 * n + 4 * e,
        "csr_rows": 8 * e + 4 * e + 8 * n,                         # links in; targets + offsets out
    }


def cpp_multi_check(fasta: Path, world: int) -> dict:
    """swarm -d 1 with SWARM_AMD_DEVICES=0..world-1 (swa_multi_*, RCCL) against the same run on GPU 0: -o md5-equal, the
    exchange reported as RCCL, wall times and the library's own milestones (SWARM_AMD_TIMING)."""
    exe = ROOT / "swarm_amd" / "bin" / "swarm"
    res = {}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            runs = {"one_gpu": {}, "all_gpus": {"SWARM_AMD_DEVICES": ",".join(str(i) for i in range(world)), "SWARM_AMD_MULTI_REPORT": "1"}}
            md5 = {}
            for tag, env in runs.items():
                t0 = time.perf_counter()
                r = subprocess.run([str(exe), "-d", "1", "-o", f"{tmp}/{tag}.o", "-l", "/dev/null", str(fasta)], capture_output=True, text=True,
                                   env=dict(os.environ, SWARM_AMD_TIMING="1", **env), timeout=150)
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    return {"error": f"{tag}: exit {r.returncode}: {r.stderr[-400:]}"}
                md5[tag] = md5_of(f"{tmp}/{tag}.o")
                res[tag] = {"seconds": round(dt, 3), "milestones": [ln for ln in r.stderr.splitlines() if ln.startswith("[t") or ln.startswith("multi:")][-12:]}
            res["output_identical"] = md5["one_gpu"] == md5["all_gpus"]
            res["exchange_is_rccl"] = any("exchange = rccl" in ln for ln in res["all_gpus"]["milestones"])
    except Exception as e:                                # (an extra must never cost the headline line)
        return {"error": f"{type(e).__name__}: {e}"}
    return res


LINE_LIMIT = 8192          # the driver's record keeps a line it can parse: r04's 17.7 KB did, r05's 21.7 KB did not — stay well below


def _short(text, limit: int = 120):
    return text if not isinstance(text, str) or len(text) <= limit else text[:limit - 3] + "..."


def _pick(src, keys, limit: int = 120) -> dict:
    return {k: _short(src[k], limit) for k in keys if isinstance(src, dict) and k in src and not isinstance(src[k], (dict, list))}


def _num(x, digits: int = 6):
    """Numbers of the side objects to a few significant digits (the headline's own keys stay exact)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, list):
        return [_num(v, digits) for v in x]
    return x


def compact_line(out: dict, detail_path: str | None = None) -> dict:
    """The ONE line the driver parses: the contract's headline keys exactly as measured, `config` =
    what names the workload, `roofline` = the dominant kernel's object + its HBM view + the step, `cpu_baseline`, and the
    other configs / the whole run as a few scalars each.  Everything else the run measured (ceilings, per-kernel traffic,
    the eight extra d=1 sets, phase tables) is `out` itself, written to the side file named under `detail`."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in head if k in out}
    cfg = out.get("config", {})
    line["config"] = _pick(cfg, ("workload", "per_gpu_queries", "db_amplicons", "step", "route", "sharding", "build", "neighbour_links",
                                 "exchange", "owned_links_rank0"), 125)
    r = out.get("roofline", {})
    roof = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "algorithmic_bytes_per_launch",
                     "dominant_group", "peak_measured", "frac_of_measured_ceiling", "step_traffic_over_minimum"), 100)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):           # (the contract's keys are always there)
        roof.setdefault(k, r.get(k))
    if isinstance(r.get("hbm"), dict):
        roof["hbm"] = _num(_pick(r["hbm"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes")))
    if isinstance(r.get("step"), dict):
        roof["step"] = _num(_pick(r["step"], ("ms", "algorithmic_bytes", "hbm_bytes_measured", "minimum_bytes", "frac_of_hbm_peak")))
    if isinstance(r.get("pair_kernels"), dict):
        roof["pair_kernels"] = _num(_pick(r["pair_kernels"], ("bound", "achieved", "peak", "unit", "frac", "ms", "hbm_bytes_measured", "frac_of_measured_ceiling")))
    if isinstance(r.get("kernels"), dict):                                        # per kernel group: ms only (the rest is in the side file)
        roof["kernel_ms"] = {g: _num(v.get("ms"), 4) for g, v in r["kernels"].items() if isinstance(v, dict)}
    line["roofline"] = roof
    for k in ("cpu_baseline", "cpu_baseline_10M"):
        if isinstance(out.get(k), dict):
            line[k] = _num(_pick(out[k], ("value", "unit", "cores", "kind", "seconds", "sample"), 125))
    wr = out.get("whole_run")
    if isinstance(wr, dict):
        w = {}
        for size, rec in wr.items():
            if isinstance(rec, dict):
                w[size] = _num(_pick(rec, ("seconds", "median_s", "p95_s", "max_s", "runs", "amplicons_per_s", "output_md5_equals_reference", "error")))
        w["what"] = "swarm_amd/bin/swarm -d 1 -o, one process from start to exit, FASTA in, swarms file out"
        line["whole_run"] = w
    if isinstance(out.get("first_step_ms"), dict):
        line["first_step_ms"] = _num(_pick(out["first_step_ms"], ("first_step_ms", "repeated_step_ms", "error")))
    if isinstance(out.get("host_seam_ms"), dict):
        line["host_seam_ms"] = _num(_pick(out["host_seam_ms"], ("upload_ms", "index_build_ms", "network_incl_download_ms", "total_ms", "amplicons_per_s", "error")))
    c1 = cfg.get("configs1")
    if isinstance(c1, dict):
        line["configs1"] = _num(_pick(c1, ("workload", "value", "unit", "ms_per_step", "neighbour_links", "error")))
    for k in ("configs2", "configs3"):
        c = out.get(k) if isinstance(out.get(k), dict) else cfg.get(k)
        if isinstance(c, dict):
            rec = _num(_pick(c, ("workload", "pipeline_total_s", "value", "unit", "counters_equal_reference_log", "qgram_comparisons_per_s",
                                 "aligned_pairs_per_s", "clustering_seconds", "full_matrix_equivalent_cells_per_s", "banded_cells_per_s", "error"), 100))
            if isinstance(c.get("cpu_baseline"), dict):
                rec["cpu_baseline"] = _num(_pick(c["cpu_baseline"], ("value", "unit", "cores", "kind", "seconds", "sample", "error"), 125))
            line[k] = rec
    for k in ("simulated", "sharded_csr_equals_whole"):
        if k in out:
            line[k] = _short(out[k], 160)
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:           # cannot happen with the keys above; if it ever does, the contract's keys survive
        for k in ("host_seam_ms", "first_step_ms", "configs1", "whole_run", "configs3", "configs2", "cpu_baseline_10M"):
            line.pop(k, None)
            if len(json.dumps(line)) < LINE_LIMIT:
                break
    return line


def emit(out: dict) -> None:
    """Side file with everything + the one compact line (the LAST thing on stdout)."""
    detail = os.environ.get("SWA_BENCH_DETAIL", str(ROOT / "bench_detail.json"))
    try:
        Path(detail).write_text(json.dumps(out, indent=1) + "\n")
        shown = os.path.relpath(detail, ROOT) if detail.startswith(str(ROOT)) else detail
    except OSError as e:
        print(f"bench.py: could not write {detail}: {e}", file=sys.stderr)
        shown = None
    print(json.dumps(compact_line(out, shown)), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--per-gpu", type=int, default=10_000_000, help="amplicons queried per GPU")
    ap.add_argument("--length", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", default="all", help="comma list of the config.* measurements to run after the timed region (default: all)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline step: no cpu_baseline, no configs1/2/3, no whole-run / seam timings, no PMC traffic")
    ap.add_argument("--no-configs1", action="store_true",
                    help="skip the extra configs[1] (1 M x 150) measurement reported under config.configs1 (N = 1 only)")
    ap.add_argument("--whole-runs", type=int, default=24, help="runs of the command line on the headline's set behind whole_run's median / p95")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="development aid: run rank 0's share of an N-GPU job on this one GPU (no collectives); "
                         "the JSON line is marked simulated and is not a result")
    ap.add_argument("--build", default="records", choices=["records", "routed", "streamed"],
                    help="N > 1, owned sharding: 'records' = every rank keys its own slice and the finished KEY RECORDS travel all-to-all "
                         "to the owners, whose build starts at the partition (swa_d1_route_slice_records + swa_d1_index_build_records); "
                         "'routed' = the ids travel and the owners key what they receive (rounds 4-5: swa_d1_route_slice + "
                         "swa_d1_index_build_routed); 'streamed' = every rank walks the whole replicated database for the keys it owns")
    ap.add_argument("--shard", default="owned", choices=["owned", "range"],
                    help="N > 1: 'owned' = every rank serves the anchor groups it owns (swa_d1_set_ownership) and the "
                         "links are exchanged all-to-all by seed range; 'range' = every rank answers its contiguous query slice "
                         "against structures indexed for that slice")
    ap.add_argument("--dev-backend", default="nccl", choices=["nccl", "gloo"],
                    help="development aid: 'gloo' runs the N>1 flow with every rank on GPU 0 (collectives staged "
                         "through the host), to exercise the sharded step on a one-GPU box; marked simulated")
    args = ap.parse_args()
    if args.no_extras:
        args.no_cpu_baseline = args.no_configs1 = True

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    one_gpu = world > 1 and args.dev_backend == "gloo"
    device_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from swarm_amd import Context, HostDb, sharding

    sim_world = args.simulate_world if world == 1 and args.simulate_world > 1 else 0
    n_total = args.per_gpu * (sim_world or world)
    fasta = Path(tempfile.gettempdir()) / f"swa_bench_{n_total}x{args.length}_s{args.seed}.fa"
    if local_rank == 0:
        gen_fasta(n_total, args.length, args.seed)
    if world > 1:
        dist.barrier()
    hdb = HostDb(fasta)
    assert hdb.n == n_total

    # database resident in HBM before the timed region (torch owns the memory; the library adopts it)
    def to_dev(a: np.ndarray, as_dtype) -> "torch.Tensor":
        return torch.from_numpy(np.ascontiguousarray(a).view(as_dtype)).to(dev)

    seqs_pad = np.concatenate([hdb.seqs, np.zeros(2, dtype=np.uint64)])
    t_seqs = to_dev(seqs_pad, np.int64)
    t_off = to_dev(hdb.seq_off, np.int64)
    t_len = to_dev(hdb.seqlen, np.int32)
    t_ab = to_dev(hdb.abundance, np.int64)
    stream = torch.cuda.current_stream(dev)
    ctx = Context(device_index, stream.cuda_stream)
    ctx.attach_db(t_seqs, t_off, t_len, t_ab, hdb.longest)
    ctx.timing_enable(True)

    parts = sharding.partition_even(n_total, sim_world or world)
    first, count = parts[rank]
    counts = [c for _, c in parts]
    owned = (sim_world or world) > 1 and args.shard == "owned"
    if owned:
        # this rank finds the links inside the anchor groups it owns, whichever amplicons they belong to
        ctx.d1_set_ownership(rank, sim_world or world)
        q_first, q_count = 0, n_total
    else:
        q_first, q_count = first, count
    cap = 8 * count
    if owned and (world > 1 or sim_world):
        d_links = torch.zeros(cap, dtype=torch.int64, device=dev)   # flat list: source << 32 | target
    else:
        d_offsets = torch.zeros(q_count + 1, dtype=torch.int64, device=dev)
        d_nb = torch.zeros(cap, dtype=torch.int32, device=dev)

    kernel_ms = []
    hits_seen = [0]
    gathered = [None]

    dup_flag = torch.zeros(1, dtype=torch.int32, device=dev)

    # routed index build (owned sharding): a rank keys only its own slice, the ids travel all-to-all to the owners of
    # their keys, every rank builds its indexes from what it received — no rank walks the whole database
    routed = owned and args.build in ("routed", "records")
    by_records = routed and args.build == "records"
    if routed:
        w_all = sim_world or world
        route_cap = 3 * count // (2 * w_all) + 1024
        d_route = torch.zeros(2 * w_all * route_cap, dtype=torch.int64 if by_records else torch.int32, device=dev)
        d_route_counts = torch.zeros(2 * w_all + 1, dtype=torch.int32, device=dev)
        sim_lists = None
        if sim_world and by_records:
            inbox = ([], [])
            for r, (f_r, c_r) in enumerate(parts):
                ctx.d1_route_slice_records(f_r, c_r, w_all, d_route, route_cap, d_route_counts)
                cts = d_route_counts.tolist()
                assert cts[2 * w_all] == 0
                for index in range(2):
                    k = index * w_all + rank
                    inbox[index].append(d_route[k * route_cap: k * route_cap + cts[k]].clone())
            sim_lists = (torch.cat(inbox[0]).contiguous(), torch.cat(inbox[1]).contiguous())
        elif sim_world:
            # one GPU playing rank 0 of N: what the other ranks would send does not change from step to step — made once,
            # outside the timed region; the timed step routes rank 0's own slice and builds from the lists
            inbox = ([], [])
            for r, (f_r, c_r) in enumerate(parts):
                ctx.d1_route_slice(f_r, c_r, w_all, d_route, route_cap, d_route_counts)
                cts = d_route_counts.tolist()
                assert cts[2 * w_all] == 0
                for index in range(2):
                    k = index * w_all + rank
                    inbox[index].append(d_route[k * route_cap: k * route_cap + cts[k]].clone())
            sim_lists = (torch.cat(inbox[0]).contiguous(), torch.cat(inbox[1]).contiguous())

    def step(record: bool) -> None:
        # every rank checks its own slice for duplicate sequences; the flags are OR-ed below
        if by_records:
            ctx.d1_route_slice_records(first, count, sim_world or world, d_route, route_cap, d_route_counts)
            rec_p, rec_s = sim_lists if sim_world else sharding.exchange_routed_records(d_route, d_route_counts, route_cap)
            torch.cuda.current_stream(dev).synchronize()      # (the lists are torch's work; the context runs on its own stream)
            dup = ctx.d1_index_build_records(rec_p, rec_s)
        elif routed:
            ctx.d1_route_slice(first, count, sim_world or world, d_route, route_cap, d_route_counts)
            ids_p, ids_s = sim_lists if sim_world else sharding.exchange_routed_ids(d_route, d_route_counts, route_cap)
            torch.cuda.current_stream(dev).synchronize()      # (the lists are torch's work; the context runs on its own stream)
            dup = ctx.d1_index_build_routed(ids_p, int(ids_p.numel()), ids_s, int(ids_s.numel()))
        else:
            dup = ctx.d1_index_build(first, count)
        # identical sequences are reported by whichever call meets them (include/swarm_amd.h): the index build when it builds a
        # table, else the network call's prefix pass (SwaError SWA_E_DUPLICATES); the ranks' findings are OR-ed AFTER both,
        # so that no rank leaves the others alone in a collective
        from swarm_amd.capi import SWA_E_DUPLICATES, SwaError
        total = 0
        try:
            if owned and (world > 1 or sim_world):
                # (a rank of an ownership-sharded job hands its links on as a flat list: no CSR over every source of the job)
                total = ctx.d1_network_edges_device(d_links, cap, False, q_first, q_count)
            else:
                total = ctx.d1_network_device(d_offsets, d_nb, cap, False, q_first, q_count)
        except SwaError as e:
            if e.code != SWA_E_DUPLICATES:
                raise
            dup = True
        if world > 1:
            dup_flag.fill_(1 if dup else 0)
            dist.all_reduce(dup_flag, op=dist.ReduceOp.MAX)
            dup = bool(dup_flag.item())
        assert not dup, "identical sequences in the synthetic set"
        hits_seen[0] = total
        if world > 1:
            if owned:
                # the links travel all-to-all by seed range and become this rank's slice of the CSR
                l_off, l_nb = sharding.exchange_owned_links(d_links[:total], counts)
            else:
                l_off, l_nb = d_offsets, d_nb[:total]
            # exchange step named by the north star: all-gather hit counts, then row offsets and
            # hit lists padded to the largest slice, so every rank holds the whole CSR
            gathered[0] = sharding.allgather_csr(l_off, l_nb, int(l_nb.numel()), counts)
        if record:
            kernel_ms.append(ctx.timing_read()[3])

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    timings = ctx.timing_read()
    sharded_ok = None
    if one_gpu and rank == 0:
        # development aid only: the gathered CSR against the whole network computed by this rank alone
        g_off, g_nb = gathered[0]
        w_off = torch.zeros(n_total + 1, dtype=torch.int64, device=dev)
        w_nb = torch.zeros(8 * n_total, dtype=torch.int32, device=dev)
        ctx.d1_set_ownership(0, 1)
        assert not ctx.d1_index_build()
        w_total = ctx.d1_network_device(w_off, w_nb, 8 * n_total, False, 0, n_total)
        sharded_ok = bool(w_total == g_nb.numel() and torch.equal(w_off, g_off) and torch.equal(w_nb[:w_total], g_nb))
    if one_gpu:
        dist.barrier()
    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        value = n_total * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        # the step's kernels: the HIP-event windows of its phases (table route: hashes / table only when built, duplicate
        # check, anchor indexes, pair kernels, CSR; streaming route: slot 7 = keys + partition + groups, slot 4 = link
        # partition + CSR rows) and, for the streaming route, of its kernel groups
        step_kernel_ms = float(timings[0] + timings[1] + timings[2] + timings[7] + k_ms + timings[4])
        st = ctx.timing_read_stream()
        streaming = st[0] > 0.0
        group_ms = {"keys": st[0], "partition_keys": st[1], "groups": st[2], "pairs0": st[3], "pairs1": st[4], "partition_links": st[5], "csr_rows": st[6]}
        if owned:   # this rank's share of the probes: the groups it owns, about 1 / world of everything
            abytes = algorithmic_bytes(hdb.seqlen, 0) / (sim_world or world) + 4.0 * hits_seen[0]
        else:
            abytes = algorithmic_bytes(hdb.seqlen[first:first + count], hits_seen[0])
        # levels of the two partitions (the library's arithmetic: buckets of <= 10 240 key records, up to 10 bits a level;
        # 2^8 sources per link bucket, 9 bits a level)
        bits = 1
        while (count >> bits) > 10240:
            bits += 1
        per_level = 10
        nbits = max(1, int(np.ceil(np.log2(max(2, q_count)))))
        model = step_byte_model(count, hits_seen[0], (bits + per_level - 1) // per_level, (max(1, nbits - min(8, nbits - 1)) + 8) // 9)
        kernels = {}
        for g, ms in group_ms.items():
            if ms > 0.0:
                kernels[g] = {"ms": ms, "algorithmic_bytes": model[g], "GB/s": model[g] / (ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": model[g] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the dominant KERNEL: among the groups that are one kernel (a partition is 6 / 12 short launches, the largest of them 0.21 /
        # 0.08 ms at 10 M — profiles/r04/*kernel_stats*.csv; both partitions stay in `kernels` and in `step`)
        single = [g for g in ("keys", "groups", "pairs0", "pairs1", "csr_rows") if g in kernels]
        dominant = max(single, key=lambda g: kernels[g]["ms"]) if single else None
        largest_group = max(kernels, key=lambda g: kernels[g]["ms"]) if kernels else None
        names = {"keys": "k_keys", "partition_keys": "k_part_hist / k_flat_* / k_part_scatter over the key records", "groups": "k_group1",
                 "partition_links": "k_part_hist / k_flat_* / k_part_scatter over the links",
                 "pairs0": "k_d1_group_pairs<0> (prefix groups)", "pairs1": "k_d1_group_pairs<1> (suffix groups)", "csr_rows": "k_csr_bucket"}
        if streaming and dominant is not None:
            dk = kernels[dominant]
            roof = {"bound": "hbm", "kernel": names[dominant] + ": the kernel with the largest share of the step's time",
                    "achieved": dk["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dk["frac_of_hbm_peak"], "traffic": None,
                    "algorithmic_bytes_per_launch": dk["algorithmic_bytes"], "avg_kernel_ms": dk["ms"],
                    "dominant_group": dominant, "largest_group_including_multi_launch": largest_group,
                    "largest_group_note": "partition_keys / partition_links are 6 / 12 short launches each (their largest single launch is shorter "
                                          "than the dominant kernel); they are graded in `kernels` and in `step`",
                    "kernels": kernels,
                    "step": {"ms": step_kernel_ms, "algorithmic_bytes": sum(model.values()),
                             "GB/s": sum(model.values()) / (step_kernel_ms * 1e-3) / 1e9,
                             "frac_of_hbm_peak": sum(model.values()) / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    "reference_equivalent_rate": {"bytes_per_step": abytes, "GB/s": abytes / (step_kernel_ms * 1e-3) / 1e9,
                                                  "note": "SURVEY.md 8(d) prices the REFERENCE's probing loop (one 8-byte membership word per microvariant, "
                                                          "8.2 KB per amplicon); the pair kernels never materialise microvariants, so this is an "
                                                          "equivalent rate, not a roofline fraction"}}
        else:
            achieved = abytes / (step_kernel_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "the d=1 step of round 2's table route as one kernel group", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": abytes,
                    "avg_kernel_ms": step_kernel_ms, "frac_note": "SURVEY.md 8(d) bytes (the reference's probing loop), not a bound for this route"}
        out = {
            "metric": "amplicons/sec clustered (d=1)",
            "value": value,
            "unit": "amplicons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"{n_total} synthetic amplicons x {args.length} bp, d=1 ({args.per_gpu} per GPU; the metric's own size; configs[1]: configs1)",
                "per_gpu_queries": count,
                "db_amplicons": n_total,
                "step": "swa_d1_index_build + swa_d1_network_device (B1 seam), db and CSR resident in HBM"
                        + ("; ownership by anchor group, links exchanged all-to-all by seed range, RCCL all-gather of CSR slices"
                           if world > 1 and owned else "; RCCL all-gather of CSR slices" if world > 1 else ""),
                "build": ("key records routed to their owners (swa_d1_route_slice_records, all-to-all, swa_d1_index_build_records)" if by_records else
                          "ids routed to their owners (swa_d1_route_slice, all-to-all, swa_d1_index_build_routed)" if routed else "local"),
                "route": "streaming (d1_stream.inc)" if streaming else "table (round 2)",
                "sharding": ("owned" if owned else "range") if (sim_world or world) > 1 else "none",
                "neighbour_links": int(hits_seen[0]),
                "phase_ms": {"seqhash": timings[0], "table_bloom_build": timings[1], "dup_check": timings[2],
                             "anchor_index_build": timings[7], "network_kernels": k_ms, "csr": timings[4]},
                "kernel_group_ms": group_ms if streaming else None,
            },
            "roofline": roof,
        }
        run_cpp_multi = False
        extras = world == 1 and not sim_world and not args.no_extras
        if world == 1 and not sim_world and not args.no_configs1:
            # BASELINE.json configs[1] (1 M x 150, d=1): the same step at that size, same run
            out["config"]["configs1"] = extra_measurement(torch, dev, device_index, args, 1_000_000, 10)
        if world > 1 and not one_gpu and not args.no_extras:
            # What the command line ships for several GPUs is swa_multi_* (multi.hip: one rank + host thread per GPU inside
            # ONE process, routed index build with grouped ncclSend / ncclRecv, link lists gathered with RCCL) — a different
            # driver of the same C entry points than the torch.distributed ranks timed above.  Run it here, on the same
            # GPUs and the same database, as a subprocess (the other ranks wait at the barrier below): its output must be
            # byte-identical to one GPU's, and it must really have used RCCL.
            # ... AFTER the line is printed (a check that has never run on several real GPUs must not be able to cost the headline:
            # it goes to the side file and to stderr, with its own short timeouts) — see the end of main
            run_cpp_multi = True
        if one_gpu:
            out["simulated"] = f"{world} ranks sharing GPU 0 over gloo: exercises the sharded step, not a result"
            out["sharded_csr_equals_whole"] = sharded_ok
        if sim_world:
            out["simulated"] = f"rank 0 of {sim_world}, no collectives: value counts all {n_total} amplicons as if every rank finished in this time"
        if extras:
            ctx.close()                                   # the headline context's buffers are not needed any more
            t_seqs = t_off = t_len = t_ab = d_offsets = d_nb = None
            torch.cuda.empty_cache()
            sample_n = min(n_total, 1_000_000)
            ref_md5 = None
            if not args.no_cpu_baseline:
                # bounded sample: the reference needs ~25 s per run on the 10 M set; 1 M keeps the three
                # thread settings it is tried with inside the 10-30 s budget
                out["cpu_baseline"] = cpu_baseline(gen_fasta(sample_n, args.length, args.seed), sample_n, args.length, args.seed)
                ref_md5 = out["cpu_baseline"].pop("output_md5", None)
            if not args.no_cpu_baseline and n_total > sample_n and (ROOT / "oracle" / "_ref" / "swarm").exists():
                # the reference on the metric's own set, once (about half a minute on 16 threads)
                t0 = time.perf_counter()
                subprocess.run([str(ROOT / "oracle" / "_ref" / "swarm"), "-d", "1", "-t", "16", "-o", "/dev/null", "-l", "/dev/null", str(fasta)], check=True)
                dt = time.perf_counter() - t0
                out["cpu_baseline_10M"] = {"value": n_total / dt, "unit": "amplicons/s", "cores": 16, "kind": "reference", "seconds": round(dt, 2),
                                           "sample": f"unmodified reference swarm 3.1.6 -d 1 -t 16, whole run on the {n_total} x {args.length} bp set of the headline"}
            for name, fn in (("skewed", lambda: extra_measurement(torch, dev, device_index, args, n_total, 5, 40)),
                             # the conserved-flank cliff: flanks that swallow a 64-nt window whole, and a
                             # V4-like set — 250 nt, 60 % of the positions conserved across centroids
                             # (200 nt: with 70 + 70 conserved of 150 the 10-nt cores of 200 000 centroids collide — identical sequences)
                             ("skewed_70", lambda: extra_measurement(torch, dev, device_index, argparse.Namespace(**{**vars(args), "length": 200}), n_total, 5, 70)),
                             ("v4_like", lambda: extra_measurement(torch, dev, device_index, argparse.Namespace(**{**vars(args), "length": 250}), n_total, 5, 0, 0.0, False, 60)),
                             ("heavy_tail", lambda: extra_measurement(torch, dev, device_index, args, n_total, 5, 0, 0.1)),
                             # 400-bp amplicons: the pair route with 13-word records (128-byte lines), 1 M x 400, d = 1
                             ("d1_x400", lambda: extra_measurement(torch, dev, device_index, argparse.Namespace(**{**vars(args), "length": 400}), 1_000_000, 5)),
                             ("d1_x460", lambda: extra_measurement(torch, dev, device_index, argparse.Namespace(**{**vars(args), "length": 460}), 1_000_000, 5)),
                             ("mixed_lengths", lambda: extra_measurement(torch, dev, device_index, args, n_total, 5, 0, 0.0, True)),
                             ("host_seam_ms", lambda: host_seam(args, n_total)),
                             ("first_step_ms", lambda: first_step(torch, dev, device_index, args, n_total)),
                             ("whole_run", lambda: whole_run(args, n_total, ref_md5, sample_n)),
                             ("configs2", lambda: config2_fastidious(args, args.per_gpu)),
                             ("configs3", lambda: config3_dn(args, 1_000_000, 400, 3))):
                if args.extras != "all" and name not in args.extras.split(","):
                    continue
                try:
                    out["config"][name] = fn()
                except Exception as e:                    # an extra must never cost the headline line
                    out["config"][name] = {"error": f"{type(e).__name__}: {e}"}
            ceil = measured_ceilings()
            if ceil is not None:
                out["roofline"]["ceilings"] = ceil
            t = measured_pmc(args)
            if t is not None and "kernels" in out["roofline"]:
                per = t["per_kernel_group"]
                out["roofline"]["traffic_detail"] = t
                dom = out["roofline"].get("dominant_group") or max(out["roofline"]["kernels"], key=lambda g: out["roofline"]["kernels"][g]["ms"], default=None)
                if "partition" in per:                          # (split over the two partitions by their algorithmic bytes)
                    kk = out["roofline"]["kernels"]
                    both = sum(kk[g]["algorithmic_bytes"] for g in ("partition_keys", "partition_links") if g in kk)
                    for g in ("partition_keys", "partition_links"):
                        if g in kk:
                            per[g] = {k: v * kk[g]["algorithmic_bytes"] / both for k, v in per["partition"].items()}
                if dom in per:
                    out["roofline"]["traffic"] = per[dom]["hbm_bytes"]
                for g, rec in out["roofline"]["kernels"].items():
                    if g in per:
                        rec["hbm_bytes_measured"] = per[g]["hbm_bytes"]
                        rec["traffic_over_algorithmic"] = per[g]["hbm_bytes"] / rec["algorithmic_bytes"]
                        if per[g]["valu_wave_instructions"] > 0:
                            rec["valu_wave_instructions"] = per[g]["valu_wave_instructions"]
                            rec["valu_wave_instructions_per_s"] = per[g]["valu_wave_instructions"] / (rec["ms"] * 1e-3)
                # the fitting ceiling of every group: HBM streaming for the streaming kernels (the copy rate this box reaches),
                # VALU issue for the pair kernels (and their line fetches against the random-line rate)
                if ceil is not None:
                    copy = ceil.get("stream_copy", {}).get("rate")
                    # (two instruction mixes are measured: the pair test's own v_ffbl / v_min and plain full-rate integer ones; the
                    # ceiling is the higher rate)
                    valu = max((ceil.get(k, {}).get("rate") or 0.0) for k in ("valu_3op", "valu_simple")) or None
                    lines = ceil.get("gather64", {}).get("rate")
                    for g, rec in out["roofline"]["kernels"].items():
                        if g.startswith("pairs"):
                            if valu and "valu_wave_instructions_per_s" in rec:
                                rec["frac_of_valu_ceiling"] = rec["valu_wave_instructions_per_s"] / valu
                            if lines:
                                rec["frac_of_random_line_ceiling"] = (count / (rec["ms"] * 1e-3)) / lines
                            rec["bound"] = "VALU issue"
                        elif copy:
                            rec["frac_of_measured_copy_rate"] = rec["GB/s"] / copy
                            rec["bound"] = "HBM streaming"
                    out["roofline"]["frac_calibrated"] = out["roofline"]["kernels"][dom].get("frac_of_valu_ceiling",
                                                                                            out["roofline"]["kernels"][dom].get("frac_of_measured_copy_rate"))
                    out["roofline"]["frac_calibrated_note"] = ("the dominant kernel group against the ceiling that fits it, measured on this box in this run "
                                                               "(tools/ubench_lines): VALU wave-instructions/s for the pair kernels, float4 copy rate for the streaming kernels")
                out["roofline"]["step"]["hbm_bytes_measured"] = t["hbm_bytes_per_step"]
                out["roofline"]["step"]["traffic_over_algorithmic"] = t["hbm_bytes_per_step"] / out["roofline"]["step"]["algorithmic_bytes"]
                # what the PROBLEM needs: every line read once, the CSR written once
                minimum = 64.0 * count + 12.0 * hits_seen[0] + 8.0 * count
                out["roofline"]["step"]["minimum_bytes"] = minimum
                out["roofline"]["step_traffic_over_minimum"] = t["hbm_bytes_per_step"] / minimum
                out["roofline"]["step_traffic_over_minimum_note"] = "measured HBM bytes of one step / (64 n lines + 12 e CSR targets and link reads + 8 n offsets)"
                # The headline object says what BINDS the dominant kernel.  A pair kernel is bound by VALU
                # issue: achieved = its VALU wave-instructions a second (SQ_INSTS_VALU / its HIP-event duration), peak = the
                # guide's issue rate; the HBM view of the same kernel — on COUNTER bytes — stays beside it under `hbm`.
                rec = out["roofline"]["kernels"].get(dom) if dom else None
                if rec is not None and "hbm_bytes_measured" in rec:
                    counter_gbs = rec["hbm_bytes_measured"] / (rec["ms"] * 1e-3) / 1e9
                    hbm_view = {"bound": "hbm", "achieved": counter_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": counter_gbs / HBM_PEAK_GBS,
                                "traffic": rec["hbm_bytes_measured"], "bytes": "FETCH_SIZE + WRITE_SIZE of this kernel, corrected as the guide prescribes",
                                "algorithmic_bytes": rec["algorithmic_bytes"], "algorithmic_GB/s": rec["GB/s"], "algorithmic_frac": rec["frac_of_hbm_peak"]}
                    if dom.startswith("pairs") and rec.get("valu_wave_instructions_per_s"):
                        rate = rec["valu_wave_instructions_per_s"]
                        out["roofline"].update({"bound": "valu", "achieved": rate, "peak": VALU_PEAK_WAVE_INSTR_S, "unit": "wave-instr/s",
                                                "frac": rate / VALU_PEAK_WAVE_INSTR_S,
                                                "valu_wave_instructions_per_launch": rec["valu_wave_instructions"],
                                                "bound_note": "integer VALU issue: the pair test is v_xor / v_ffbl / v_ffbh / v_min chains over packed words in "
                                                              "registers; its line fetches (hbm.frac) leave the memory system mostly idle"})
                        if ceil is not None:
                            meas = max((ceil.get(k, {}).get("rate") or 0.0) for k in ("valu_3op", "valu_simple")) or None
                            if meas:
                                out["roofline"]["peak_measured"] = meas
                                out["roofline"]["frac_of_measured_ceiling"] = rate / meas
                        out["roofline"]["hbm"] = hbm_view
                    else:
                        out["roofline"].update({"achieved": counter_gbs, "frac": counter_gbs / HBM_PEAK_GBS,
                                                "achieved_note": "counter bytes of the dominant kernel / its duration"})
                        if dom == "groups":
                            out["roofline"]["bound_note"] = ("k_group1 is bound by the NUMBER of instructions its sixteen waves issue against LDS-resident tables "
                                                             "(DESIGN 3.2: 32 k instructions a wave for ~10 records a lane), neither by HBM nor by VALU throughput; "
                                                             "the HBM fraction is what the contract asks for, the pair kernels - one per cent shorter on this box - are "
                                                             "graded by VALU issue under `pair_kernels`")
                        out["roofline"]["hbm"] = hbm_view
                # the pair kernels by the resource that binds them, whichever kernel is the longest on this box (k_group1 and the
                # two pair passes are within two per cent of one another)
                pk = [out["roofline"]["kernels"][g] for g in ("pairs0", "pairs1") if g in out["roofline"]["kernels"] and out["roofline"]["kernels"][g].get("valu_wave_instructions_per_s")]
                if pk:
                    rate = sum(k["valu_wave_instructions"] for k in pk) / sum(k["ms"] * 1e-3 for k in pk)
                    out["roofline"]["pair_kernels"] = {"bound": "valu", "achieved": rate, "peak": VALU_PEAK_WAVE_INSTR_S, "unit": "wave-instr/s",
                                                       "frac": rate / VALU_PEAK_WAVE_INSTR_S, "ms": sum(k["ms"] for k in pk),
                                                       "hbm_bytes_measured": sum(k.get("hbm_bytes_measured", 0.0) for k in pk)}
                    if ceil is not None:
                        meas = max((ceil.get(k, {}).get("rate") or 0.0) for k in ("valu_3op", "valu_simple")) or None
                        if meas:
                            out["roofline"]["pair_kernels"]["frac_of_measured_ceiling"] = rate / meas
            # what the driver's record keeps is the top level of this line: the numbers a reader of BENCH_rNN.json needs
            # beside `value` live there, not under config
            for name in ("whole_run", "host_seam_ms", "first_step_ms"):
                if name in out["config"]:
                    out[name] = out["config"].pop(name)
            c3 = out["config"].get("configs3")
            if isinstance(c3, dict) and "qgram_comparisons_per_s" in c3:
                out["configs3"] = {k: c3[k] for k in ("workload", "value", "unit", "qgram_comparisons_per_s", "aligned_pairs_per_s", "clustering_seconds",
                                                      "full_matrix_equivalent_cells_per_s", "banded_cells_per_s", "cpu_baseline") if k in c3}
            c2 = out["config"].get("configs2")
            if isinstance(c2, dict) and "pipeline_total_s" in c2:
                out["configs2"] = {k: c2[k] for k in ("workload", "pipeline_total_s", "value", "unit", "counters_equal_reference_log", "cpu_baseline") if k in c2}
        elif world == 1 and not sim_world and not args.no_cpu_baseline:
            sample_n = min(n_total, 1_000_000)
            out["cpu_baseline"] = cpu_baseline(gen_fasta(sample_n, args.length, args.seed), sample_n, args.length, args.seed)
            out["cpu_baseline"].pop("output_md5", None)
        emit(out)
        if run_cpp_multi:
            out["config"]["cpp_multi"] = cpp_multi_check(fasta, world)
            print("cpp_multi: " + json.dumps(out["config"]["cpp_multi"])[:1500], file=sys.stderr, flush=True)
            try:
                Path(os.environ.get("SWA_BENCH_DETAIL", str(ROOT / "bench_detail.json"))).write_text(json.dumps(out, indent=1) + "\n")
            except OSError:
                pass
    if not (rank == 0 and world == 1 and not sim_world and not args.no_extras):
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
