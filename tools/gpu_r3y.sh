#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3y
mkdir -p $O
cd $R
$R/tools/ubench_lines $O/ubench_quick.json quick > /dev/null 2>&1; python - $O/ubench_quick.json <<'PY'
import json, sys
for r in json.load(open(sys.argv[1]))["results"]:
    if r["test"].startswith("valu"): print(r["test"], "%.3e" % r["rate"])
PY
( time timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_fastidious_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
for n in (10_000_000, 1_000_000):
    a = argparse.Namespace(length=150, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, n, 10)
    print(n, round(r["ms_per_step"], 4), {k: round(v, 3) for k, v in r["kernel_group_ms"].items()}, r["neighbour_links"], flush=True)
a = argparse.Namespace(length=150, seed=1, per_gpu=10_000_000)
r = bench.config2_fastidious(a, 10_000_000)
print("fastidious", r["fastidious_kernels_ms"])
PY
tail -3 $O/x.err
