#!/usr/bin/env python3
"""Phase timings of the whole pipeline for the BASELINE.json configs beyond the one bench.py
reports (run on the GPU box through gpurun; results go to gpurun_out/ and, once curated, to
profiles/).

    python tools/bench_configs.py d1 --n 10000000 --length 150 --fastidious [--reference]
    python tools/bench_configs.py dn --n 200000 --length 400 -d 3 [--reference]
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

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class Timer:
    def __init__(self):
        self.t = {}

    def run(self, name, fn):
        t0 = time.perf_counter()
        out = fn()
        self.t[name] = round(time.perf_counter() - t0, 4)
        return out


def gen(n, length, seed, edits, light):
    import bench
    return bench.gen_fasta(n, length, seed, edits, float(light))


def reference(fa, args, threads):
    ref = ROOT / "oracle" / "_ref" / "swarm"
    if not ref.exists():
        return None
    out = Path(tempfile.gettempdir()) / "swa_ref.out"
    t0 = time.perf_counter()
    subprocess.run([str(ref)] + args + ["-t", str(threads), "-o", str(out), "-l", "/dev/null", str(fa)], check=True)
    return {"seconds": round(time.perf_counter() - t0, 2), "threads": threads, "out": str(out)}


def dn_rates(scan, seconds, length):
    """q-gram comparisons/s, aligned pairs/s and full-matrix-equivalent DP cells/s (the reference
    fills Lq x Lt cells per pair, src/search8.cc; the wavefront / banded kernels touch far fewer)."""
    return {"qgram_comparisons_per_s": scan["qgram_comparisons"] / seconds,
            "aligned_pairs_per_s": scan["aligned_pairs"] / seconds,
            "full_matrix_equivalent_cells_per_s": scan["aligned_pairs"] * length * length / seconds,
            "over": "dn_cluster_gpu_scan (host greedy loop + all launches), nominal length squared per pair"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["d1", "dn"])
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--length", type=int, default=150)
    ap.add_argument("-d", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--fastidious", action="store_true")
    ap.add_argument("--reference", action="store_true")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()

    from swarm_amd import Context, D1Clusters, DnClusters, HostDb

    T = Timer()
    res = {"mode": args.mode, "n": args.n, "length": args.length, "host_cores": os.cpu_count()}
    if args.mode == "d1":
        light = 0.3 if args.fastidious else 0.0
        fa = T.run("generate", lambda: gen(args.n, args.length, args.seed, 1, light))
        hdb = T.run("fasta_read_sort_pack", lambda: HostDb(fa))
        ctx = Context(0)
        ctx.timing_enable(True)
        T.run("upload", lambda: ctx.upload_hostdb(hdb))
        T.run("index_build", ctx.d1_index_build)
        T.run("network_resident", ctx.d1_network_resident)     # (the command line's route: the network stays in HBM)
        ms = ctx.timing_read()
        res["gpu_ms"] = {"seqhash": ms[0], "table": ms[1], "dup": ms[2], "network": ms[3], "csr": ms[4]}
        cl = T.run("clustering_gpu_plus_sums", lambda: D1Clusters.from_resident(ctx, hdb))
        if args.fastidious:
            flags, stats = T.run("light_flags", cl.light_flags)
            graft, counters = T.run("fastidious_gpu", lambda: ctx.d1_fastidious(flags, stats[2]))
            ms = ctx.timing_read()
            res["gpu_ms"]["light_pass"] = ms[5]
            res["gpu_ms"]["heavy_pass"] = ms[6]
            res["fastidious"] = {"light_swarms": stats[0], "light_amplicons": stats[1], "light_variants": int(counters[0]),
                                 "heavy_variants": int(counters[1]), "candidates": int(counters[2])}
            res["grafts"] = T.run("graft", lambda: cl.graft(graft))
        out = Path(tempfile.gettempdir()) / "swa_gpu.out"
        T.run("write_swarms", lambda: cl.write_swarms(out))
        res["summary"] = cl.summary()
        refargs = ["-d", "1"] + (["-f"] if args.fastidious else [])
    else:
        fa = T.run("generate", lambda: gen(args.n, args.length, args.seed, args.d, 0.0))
        hdb = T.run("fasta_read_sort_pack", lambda: HostDb(fa, check_duplicate_sequences=True))
        ctx = Context(0)
        T.run("upload", lambda: ctx.upload_hostdb(hdb))
        cl = T.run("dn_cluster_gpu_scan", lambda: DnClusters(ctx, hdb, args.d))
        res["scan"] = cl.scan_totals()
        out = Path(tempfile.gettempdir()) / "swa_gpu.out"
        T.run("write_swarms", lambda: cl.write_swarms(out))
        res["summary"] = cl.summary()
        refargs = ["-d", str(args.d)]
    res["seconds"] = T.t
    if "scan" in res:                                      # the three d>=2 rates, over the clustering phase
        res["rates"] = dn_rates(res["scan"], T.t["dn_cluster_gpu_scan"], args.length)
    res["total_seconds_without_generate"] = round(sum(v for k, v in T.t.items() if k != "generate"), 3)
    if args.reference:
        r = reference(fa, refargs, args.threads)
        if r:
            same = subprocess.run(["cmp", "-s", r["out"], str(out)]).returncode == 0
            r["output_identical"] = same
            if not same:                                   # leave enough behind to tell which side is off
                r["sizes"] = [os.path.getsize(r["out"]), os.path.getsize(out)]
                r["first_difference"] = subprocess.run(["cmp", r["out"], str(out)], capture_output=True, text=True).stdout.strip()
            res["reference"] = r
    print(json.dumps(res))


if __name__ == "__main__":
    main()
