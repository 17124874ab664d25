#!/usr/bin/env python3
"""tools/bench_extra.py <n> <steps> [flank] [zipf] — one of bench.py's extra step measurements on its own (development aid):
the d=1 step on n x 150 amplicons with conserved flanks / Zipf family sizes, per kernel group."""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main() -> None:
    import torch
    n, steps = int(sys.argv[1]), int(sys.argv[2])
    flank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    zipf = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    args = argparse.Namespace(length=150, seed=1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    print(json.dumps(bench.extra_measurement(torch, dev, 0, args, n, steps, flank, zipf)))


if __name__ == "__main__":
    main()
