#!/bin/bash
# round 4, lease ae: bundles per counter visit chosen by the number of bundles — 1 M, 3 M, 10 M; parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4ae; mkdir -p $O
cd $R
for n in 1000000 3000000 10000000; do
  timeout 300 python bench.py --per-gpu $n --steps 10 --warmup 3 --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r4ae/bench_$n.json").read().strip().splitlines()[-1])
    print($n, round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["config"]["kernel_group_ms"].items()})
except Exception as e: print($n, "ERR", e)
PY
done
for b in 1 4; do SWA_D1_PAIR_BATCH=$b timeout 300 python bench.py --per-gpu 3000000 --steps 10 --warmup 3 --no-extras > $O/bench_3M_b$b.json 2> $O/bench_3M_b$b.err
python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open("gpurun_out/r4ae/bench_3M_b$b.json").read().strip().splitlines()[-1])
print("3M batch", $b, round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["config"]["kernel_group_ms"].items()})
PY
done
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_lengths_gpu.py tests/test_guard_gpu.py tests/test_d1_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
