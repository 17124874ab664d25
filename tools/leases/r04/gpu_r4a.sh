#!/bin/bash
# round 4, lease a: guard + wide anchors — targeted tests, the anomaly hunt, first bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
rocm-smi --showserial > $O/serial.txt 2>&1
timeout 900 python -m pytest tests/test_guard_gpu.py tests/test_stream_gpu.py tests/test_d1_gpu.py -x -q -m gpu > $O/tests_d1.log 2>&1; echo "tests_d1 rc=$?" >> $O/status.txt
tail -5 $O/tests_d1.log
timeout 400 python tools/stress/run.py 160 200000 > $O/stress.log 2>&1; echo "stress rc=$?" >> $O/status.txt
cp gpurun_out/stress/summary.json $O/stress_summary.json 2>/dev/null
tail -30 $O/stress.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_w_auto.json 2> $O/bench_w_auto.err; echo "bench rc=$?" >> $O/status.txt
SWA_D1_ANCHOR_W=32 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_w32.json 2> $O/bench_w32.err
python - <<'PY'
import json,sys
for f in ("bench_w_auto","bench_w32"):
    try:
        d=json.loads(open(f"gpurun_out/r4a/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], {k:round(v["ms"],3) for k,v in d["roofline"]["kernels"].items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 600 python -m pytest tests/test_multi_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "not 100" > $O/tests_multi_full.log 2>&1; echo "tests_multi_full rc=$?" >> $O/status.txt
tail -5 $O/tests_multi_full.log
cat $O/status.txt
