#!/bin/bash
# round 4, lease h: k_group1 variants (records in registers x batched probing)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
for v in v80 v81 v100 v101; do
  SWARM_AMD_LIB=$R/swarm_amd/lib/libswarm_amd_$v.so timeout 200 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for v in ("v80","v81","v100","v101"):
    try:
        d=json.loads(open(f"gpurun_out/r4h/bench_{v}.json").read().strip().splitlines()[-1])
        print(v, round(d["ms_per_step"],3), {k:round(x["ms"],3) for k,x in d["roofline"]["kernels"].items()})
    except Exception as e: print(v, "ERR", e)
PY
