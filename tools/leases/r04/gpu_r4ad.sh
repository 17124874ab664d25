#!/bin/bash
# round 4, lease ad: bundles per visit of the pair kernels' work counters at 1 M (4 = the 10 M choice)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4ad; mkdir -p $O
cd $R
for b in 4 2 1; do
  SWA_D1_PAIR_BATCH=$b timeout 300 python bench.py --per-gpu 1000000 --steps 20 --warmup 3 --no-extras > $O/bench_b$b.json 2> $O/bench_b$b.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r4ad/bench_b$b.json").read().strip().splitlines()[-1])
    print("batch", $b, round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["config"]["kernel_group_ms"].items()})
except Exception as e: print($b, "ERR", e)
PY
done
