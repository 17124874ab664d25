#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3w
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
timeout 300 python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
for sweep in ("1", "0"):
  os.environ["SWA_D1_SWEEP"] = sweep
  for n in (10_000_000, 1_000_000):
    a = argparse.Namespace(length=150, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, n, 10)
    print("sweep", sweep, n, round(r["ms_per_step"], 4), {k: round(v, 3) for k, v in r["kernel_group_ms"].items()}, r["neighbour_links"], flush=True)
PY
tail -3 $O/x.err
