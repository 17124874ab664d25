#!/bin/bash
# round 4, lease g: k_group1 variants (records in registers x batched probing), 16-lane WFA groups on the d >= 2 tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
for v in v80 v81 v100 v101; do
  SWARM_AMD_LIB=$R/swarm_amd/lib/libswarm_amd_$v.so timeout 200 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for v in ("v80","v81","v100","v101"):
    try:
        d=json.loads(open(f"gpurun_out/r4g/bench_{v}.json").read().strip().splitlines()[-1])
        print(v, round(d["ms_per_step"],3), {k:round(x["ms"],3) for k,x in d["roofline"]["kernels"].items()})
    except Exception as e: print(v, "ERR", e)
PY
timeout 900 python -m pytest tests/test_dn_gpu.py tests/test_scan_gpu.py -q -m gpu -x > $O/tests_dn.log 2>&1; echo "tests_dn rc=$?" >> $O/status.txt
tail -4 $O/tests_dn.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras configs3 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4g/bench_c3.json").read().strip().splitlines()[-1])
print(json.dumps(d["config"].get("configs3"))[:1200])
PY
cat $O/status.txt
