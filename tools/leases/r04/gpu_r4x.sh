#!/bin/bash
# round 4, lease x: the fresh-process first-step hunt once more on the round's last kernels; simulated rank-0 share of 8 x 10 M
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4x; mkdir -p $O
cd $R
timeout 300 python tools/stress/run.py 64 200000 > $O/stress.log 2>&1; echo "stress rc=$?" | tee -a $O/status.txt
cp gpurun_out/stress/summary.json $O/stress_summary.json 2>/dev/null
tail -6 $O/stress.log
timeout 600 python bench.py --steps 5 --warmup 2 --simulate-world 8 --no-extras > $O/bench_sim8.json 2> $O/bench_sim8.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r4x/bench_sim8.json").read().strip().splitlines()[-1])
    print("sim8", d["ms_per_step"], d["config"].get("phase_ms"), d["config"].get("kernel_group_ms"))
except Exception as e: print("sim8 ERR", e)
PY
