#!/bin/bash
# round 4, lease af: where k_csr_bucket spends its time (the kernel cut short after each stage)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4af; mkdir -p $O
cd $R
for v in base csr1 csr2 csr3 csr4; do
  lib=$R/swarm_amd/lib/libswarm_amd_$v.so; [ $v = base ] && lib=$R/swarm_amd/lib/libswarm_amd.so
  SWARM_AMD_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r4af/bench_$v.json").read().strip().splitlines()[-1])
    print("$v", round(d["ms_per_step"],3), "csr_rows", round(d["config"]["kernel_group_ms"]["csr_rows"],4))
except Exception as e: print("$v", "ERR", e, open("gpurun_out/r4af/bench_$v.err").read()[-300:])
PY
done
