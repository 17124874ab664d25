#!/bin/bash
# round 4, lease ab: key partition in tiles of 8192 (runs of 64 bytes) against 4096
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4ab; mkdir -p $O
cd $R
for t in 4096 8192 4096 8192; do
  SWA_D1_KEY_TILE=$t timeout 300 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench_$t.json 2> $O/bench_$t.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r4ab/bench_$t.json").read().strip().splitlines()[-1])
    print($t, round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["config"]["kernel_group_ms"].items()})
except Exception as e: print($t, "ERR", e)
PY
done
SWA_D1_KEY_TILE=8192 timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_lengths_gpu.py tests/test_guard_gpu.py -m gpu -x -q > $O/tests_8192.log 2>&1; echo "tests 8192 rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests_8192.log
