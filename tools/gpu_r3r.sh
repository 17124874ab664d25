#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3r
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_fastidious_gpu.py tests/test_stream_gpu.py tests/test_d1_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
( time timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "not 100" ) > $O/tests_full.log 2>&1; grep -E "passed|failed|rror" $O/tests_full.log | tail -3
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench
for mode in ("lines", "words"):
    os.environ["SWA_FAST_PAIRS"] = mode
    a = argparse.Namespace(length=150, seed=1, per_gpu=10_000_000)
    r = bench.config2_fastidious(a, 10_000_000)
    print(mode, json.dumps({k: r[k] for k in ("fastidious_kernels_ms", "pipeline_seconds") if k in r}), flush=True)
PY
tail -3 $O/x.err
