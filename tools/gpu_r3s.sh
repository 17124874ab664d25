#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3s
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_dn_gpu.py tests/test_scan_gpu.py tests/test_cli_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
( time timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -k "not 100m" ) > $O/tests_full.log 2>&1; grep -E "passed|failed|rror" $O/tests_full.log | tail -3
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench
for mode in ("device", "host"):
    os.environ["SWARM_AMD_DN_WALK"] = mode
    a = argparse.Namespace(length=400, seed=1, per_gpu=1_000_000)
    for rep in range(2):
        r = bench.config3_dn(a, 1_000_000, 400, 3)
        print(mode, rep, json.dumps({k: r[k] for k in ("clustering_seconds", "swarms", "route", "kernel_launches") if k in r}), flush=True)
PY
tail -5 $O/x.err
