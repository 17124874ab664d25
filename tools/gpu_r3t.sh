#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3t
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_multi_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
for n in (10_000_000, 1_000_000):
    a = argparse.Namespace(length=150, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, n, 10)
    print(n, round(r["ms_per_step"], 4), {k: round(v, 3) for k, v in r["kernel_group_ms"].items()}, r["neighbour_links"], flush=True)
PY
tail -3 $O/x.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof10 -o p -- python $R/bench.py --steps 5 --warmup 2 --no-extras > $O/prof10.log 2>&1
f=$(find $O/prof10 -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_10M.csv
python - $O/kernel_stats_10M.csv <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    m = re.search(r"(k_\w+(<[^>(]*>)?)", r["Name"]); name = m.group(1) if m else r["Name"][:50]
    print(f"{name:40s} calls {int(r['Calls']):4d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f}")
PY
