#!/usr/bin/env python3
"""Rank 0's share of an N-GPU weak-scaled d=1 step, measured on ONE GPU (no collectives):
runs `bench.py --simulate-world N` for N in 1, 2, 4, 8 and both sharding schemes and prints the
phase table of DESIGN.md section 6 (markdown) plus one JSON line per run.

    python tools/scaling_share.py [--per-gpu 1000000] [--steps 5] > gpurun_out/scaling_share.md
"""
from __future__ import annotations

import argparse
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PHASES = ["seqhash", "table_bloom_build", "dup_check", "anchor_index_build", "network_kernels", "csr"]


def run(world: int, shard: str, per_gpu: int, steps: int) -> dict:
    cmd = [sys.executable, str(ROOT / "bench.py"), "--per-gpu", str(per_gpu), "--steps", str(steps), "--warmup", "1",
           "--no-cpu-baseline", "--no-configs1", "--shard", shard]
    if world > 1:
        cmd += ["--simulate-world", str(world)]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
    return json.loads([line for line in out.splitlines() if line.startswith("{")][-1])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-gpu", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--worlds", type=int, nargs="*", default=[1, 2, 4, 8])
    args = ap.parse_args()
    rows, raw = [], []
    for world in args.worlds:
        for shard in (["owned"] if world == 1 else ["owned", "range"]):
            d = run(world, shard, args.per_gpu, args.steps)
            raw.append({"world": world, "shard": d["config"]["sharding"], "ms_per_step": d["ms_per_step"],
                        "phase_ms": d["config"]["phase_ms"], "links": d["config"]["neighbour_links"]})
            ph = d["config"]["phase_ms"]
            rows.append(f"| {world} | {d['config']['sharding']} | {d['ms_per_step']:.2f} | "
                        + " | ".join(f"{ph[k]:.2f}" for k in PHASES) + " |")
    print(f"per GPU: {args.per_gpu} amplicons\n")
    print("| N | scheme | step ms | " + " | ".join(PHASES) + " |")
    print("|---|---|---|" + "---|" * len(PHASES))
    print("\n".join(rows))
    print()
    for r in raw:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
