#!/usr/bin/env python3
"""bench.py — throughput of the d=1 neighbour-finding path (seam B1) on N MI355X.

One "step" = one full pass of the hot path over the synthetic amplicon set, through the
C ABI, with the packed database already resident in HBM and the CSR left in HBM:
    swa_d1_index_build   (sequence hashes, amplicon hash table, Bloom filter, duplicate check)
    swa_d1_network_device (microvariant hashes -> Bloom -> table probe -> verify -> CSR)
and, for N > 1, a MAX all-reduce of the per-rank duplicate flags, the all-to-all merge of the
ranks' links by seed range and the RCCL all-gather of the per-rank CSR slices over xGMI.

Workload at N = 1: the size BASELINE.json's metric string names — 10 M synthetic amplicons x
150 bp, d = 1 (configs[1], 1 M x 150, is measured in the same run and reported under
config.configs1).  For N > 1 the job is weak-scaled: the database holds N x 10 M amplicons
(N = 8: 80 M, the order of configs[4]), replicated on every GPU.  Default sharding ("owned"):
rank r serves the anchor groups whose key maps to r — with all their members, so the per-group
LDS tables are built once per job, not once per rank —, which leaves it a share of the
links; they travel all-to-all by seed range, become the rank's slice of the CSR, and the slices
are all-gathered.  `--shard range` is the older scheme (rank r answers its contiguous slice).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline"     — algorithmic bytes of the dominant kernel group (the d=1 network: the anchored
                   passes k_d1_anchor + the fallback probe, which together issue one probe per
                   microvariant) / its measured average duration per step (HIP events on the
                   launch stream) vs the 8 TB/s HBM peak
  "cpu_baseline" — the unmodified reference (oracle/_ref/swarm, kind "reference") or the C
                   oracle (kind "port") timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


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


def gen_fasta(n: int, length: int, seed: int, edits: int = 1, light: float = 0.0) -> Path:
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
    fasta = Path(tempfile.gettempdir()) / f"swa_bench_{n}x{length}_s{seed}.fa"
    if fasta.exists():
        return fasta
    tool = str(gen_tool())
    tmp = fasta.with_suffix(f".tmp{os.getpid()}")
    block = 2_000_000
    if n <= block:
        subprocess.run([tool, str(n), str(length), str(seed), "1", "0", str(tmp)], check=True)
    else:
        parts, procs, at, b = [], [], 0, 0
        limit = max(1, min(32, (os.cpu_count() or 2) // 2))
        while at < n:
            size = min(block, n - at)
            part = fasta.with_suffix(f".part{b}.{os.getpid()}")
            parts.append(part)
            procs.append(subprocess.Popen([tool, str(size), str(length), str(seed * 1000 + b + 1), "1", "0", str(part), str(at)]))
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
                subprocess.run([str(ref), "-d", "1", "-t", str(threads), "-o", "/dev/null", "-l", "/dev/null",
                                str(sample)], check=True)
                dt = time.perf_counter() - t0
                tried.append(f"-t {threads}: {dt:.2f} s")
                if best is None or dt < best[1]:
                    best = (threads, dt)
            threads, dt = best
            return {"value": sample_n / dt, "unit": "amplicons/s", "cores": threads, "kind": "reference",
                    "sample": f"unmodified reference swarm 3.1.6 (oracle/_ref/swarm) -d 1, whole run (FASTA read + "
                              f"network + clustering + output) on {sample_n} x {length} bp synthetic amplicons, best "
                              f"of [{'; '.join(tried)}] on a {cores}-core host"}
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


def extra_measurement(torch, dev, device_index: int, args, n: int, steps: int) -> dict:
    """The bench step (index build + network, db and CSR resident) on n x length amplicons."""
    from swarm_amd import Context, HostDb
    hdb = HostDb(gen_fasta(n, args.length, args.seed))

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
    for it in range(1 + steps):
        if it == 1:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        assert not ctx.d1_index_build()
        total = ctx.d1_network_device(d_offsets, d_nb, cap, False, 0, hdb.n)
        if it >= 1:
            k_ms.append(ctx.timing_read()[3])
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    k = float(np.mean(k_ms))
    abytes = algorithmic_bytes(hdb.seqlen, total)
    ctx.close()
    return {"workload": f"{hdb.n} synthetic amplicons x {args.length} bp, d=1", "value": hdb.n * steps / elapsed,
            "unit": "amplicons/s", "steps": steps, "ms_per_step": 1000.0 * elapsed / steps, "network_kernels_ms": k,
            "neighbour_links": int(total), "roofline_frac": abytes / (k * 1e-3) / 1e9 / HBM_PEAK_GBS}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--per-gpu", type=int, default=10_000_000, help="amplicons queried per GPU")
    ap.add_argument("--length", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true",
                    help="skip the extra configs[1] (1 M x 150) measurement reported under config.configs1 (N = 1 only)")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="development aid: run rank 0's share of an N-GPU job on this one GPU (no collectives); "
                         "the JSON line is marked simulated and is not a result")
    ap.add_argument("--shard", default="owned", choices=["owned", "range"],
                    help="N > 1: 'owned' = every rank serves the anchor groups it owns (swa_d1_set_ownership) and the "
                         "links are exchanged all-to-all by seed range; 'range' = every rank answers its contiguous query slice "
                         "against structures indexed for that slice")
    ap.add_argument("--dev-backend", default="nccl", choices=["nccl", "gloo"],
                    help="development aid: 'gloo' runs the N>1 flow with every rank on GPU 0 (collectives staged "
                         "through the host), to exercise the sharded step on a one-GPU box; marked simulated")
    args = ap.parse_args()

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
    if owned and world > 1:
        d_links = torch.zeros(cap, dtype=torch.int64, device=dev)   # flat list: source << 32 | target
    else:
        d_offsets = torch.zeros(q_count + 1, dtype=torch.int64, device=dev)
        d_nb = torch.zeros(cap, dtype=torch.int32, device=dev)

    kernel_ms = []
    hits_seen = [0]
    gathered = [None]

    dup_flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def step(record: bool) -> None:
        # every rank checks its own slice for duplicate sequences; the flags are OR-ed below
        dup = ctx.d1_index_build(first, count)
        if world > 1:
            dup_flag.fill_(1 if dup else 0)
            dist.all_reduce(dup_flag, op=dist.ReduceOp.MAX)
            dup = bool(dup_flag.item())
        assert not dup
        if owned and world > 1:
            total = ctx.d1_network_edges_device(d_links, cap, False, q_first, q_count)
        else:
            total = ctx.d1_network_device(d_offsets, d_nb, cap, False, q_first, q_count)
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
        if owned:   # this rank's share of the probes: the groups it owns, about 1 / world of everything
            abytes = algorithmic_bytes(hdb.seqlen, 0) / (sim_world or world) + 4.0 * hits_seen[0]
        else:
            abytes = algorithmic_bytes(hdb.seqlen[first:first + count], hits_seen[0])
        achieved = abytes / (k_ms * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "d1_network_pmc.json"
        if pmc.exists():
            try:
                rec = json.loads(pmc.read_text())
                if rec.get("workload") == f"{args.per_gpu}x{args.length}_s{args.seed}":
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
                "workload": f"{n_total} synthetic amplicons x {args.length} bp, d=1 ({args.per_gpu} per GPU: the size "
                            "BASELINE.json's metric names; configs[1] under config.configs1)",
                "per_gpu_queries": count,
                "db_amplicons": n_total,
                "step": "swa_d1_index_build + swa_d1_network_device (B1 seam), db and CSR resident in HBM"
                        + ("; ownership by anchor group, links exchanged all-to-all by seed range, RCCL all-gather of CSR slices"
                           if world > 1 and owned else "; RCCL all-gather of CSR slices" if world > 1 else ""),
                "sharding": ("owned" if owned else "range") if (sim_world or world) > 1 else "none",
                "neighbour_links": int(hits_seen[0]),
                "phase_ms": {"seqhash": timings[0], "table_bloom_build": timings[1], "dup_check": timings[2],
                             "anchor_index_build": timings[7], "network_kernels": k_ms, "csr": timings[4]},
            },
            "roofline": {"bound": "hbm", "kernel": "d=1 network = k_d1_anchor<small|big> x (prefix, suffix pass) + k_d1_probe<MODE 2> "
                                   "fallback: together one probe per microvariant; duration = their sum per step", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes, "avg_kernel_ms": k_ms},
        }
        if world == 1 and not sim_world and not args.no_configs1 and not args.no_cpu_baseline:
            # BASELINE.json configs[1] (1 M x 150, d=1): the same step at that size, same run
            out["config"]["configs1"] = extra_measurement(torch, dev, device_index, args, 1_000_000, 10)
        if one_gpu:
            out["simulated"] = f"{world} ranks sharing GPU 0 over gloo: exercises the sharded step, not a result"
            out["sharded_csr_equals_whole"] = sharded_ok
        if sim_world:
            out["simulated"] = f"rank 0 of {sim_world}, no collectives: value counts all {n_total} amplicons as if every rank finished in this time"
        if world == 1 and not sim_world and not args.no_cpu_baseline:
            # bounded sample: the reference needs ~25 s per run on the 10 M set; 1 M keeps the three
            # thread settings it is tried with inside the 10-30 s budget
            sample_n = min(n_total, 1_000_000)
            out["cpu_baseline"] = cpu_baseline(gen_fasta(sample_n, args.length, args.seed), sample_n, args.length, args.seed)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
