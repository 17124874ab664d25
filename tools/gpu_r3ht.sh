#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3ht
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
a = argparse.Namespace(length=150, seed=1)
r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, 10_000_000, 5, zipf=0.1)
print("heavy_tail", round(r["ms_per_step"], 3), {k: round(v, 3) for k, v in r["kernel_group_ms"].items()}, r["neighbour_links"], flush=True)
r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, 10_000_000, 10)
print("10M", round(r["ms_per_step"], 3), {k: round(v, 3) for k, v in r["kernel_group_ms"].items()}, r["neighbour_links"], flush=True)
PY
tail -3 $O/x.err
