#!/bin/bash
# round 4, lease j: d >= 2 host timeline, simulated rank-0 share of an 8 x 10 M job, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
SWARM_AMD_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras configs3 > $O/bench_c3.json 2> $O/bench_c3.err
grep "\[dn" $O/bench_c3.err | head -20
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4j/bench_c3.json").read().strip().splitlines()[-1])
c=d["config"].get("configs3"); print({k:c[k] for k in ("clustering_seconds","gpu_kernels_ms","aligned_pairs")} if isinstance(c,dict) and "clustering_seconds" in c else c)
PY
timeout 600 python bench.py --steps 5 --warmup 2 --simulate-world 8 --no-extras > $O/bench_sim8.json 2> $O/bench_sim8.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r4j/bench_sim8.json").read().strip().splitlines()[-1])
    print("sim8", d["ms_per_step"], d["config"].get("phase_ms"), d["config"].get("kernel_group_ms"))
except Exception as e: print("sim8 ERR", e)
PY
tail -3 $O/bench_sim8.err
timeout 1500 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests_all rc=$?" >> $O/status.txt
tail -6 $O/tests_all.log
cat $O/status.txt
