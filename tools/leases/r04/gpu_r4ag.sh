#!/bin/bash
# round 4, lease ah: + rows of 33..64 by a rank sort of the wave, long rows found by ballot — bench, parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4ah; mkdir -p $O
cd $R
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open("gpurun_out/r4ah/bench_$i.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["config"]["kernel_group_ms"].items()})
PY
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail,skewed > $O/bench_ht.json 2> $O/bench_ht.err
python - <<'PY' | tee -a $O/summary.txt
import json
d=json.loads(open("gpurun_out/r4ah/bench_ht.json").read().strip().splitlines()[-1])
for k in ("heavy_tail","skewed"):
    v=d["config"].get(k); print(k, v.get("ms_per_step"), v.get("kernel_group_ms"))
PY
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_lengths_gpu.py tests/test_guard_gpu.py tests/test_d1_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
