#!/bin/bash
# round 4, lease z: largest group a workgroup takes by pairs (64 / 128 / 256) at 1 M and 10 M
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4z; mkdir -p $O
cd $R
for n in 1000000 10000000; do for big in 64 128 256; do
  SWA_D1_PAIR_BIG=$big timeout 300 python bench.py --per-gpu $n --steps 10 --warmup 2 --no-extras > $O/bench_${n}_$big.json 2> $O/bench_${n}_$big.err
done; done
python - <<'PY' | tee $O/summary.txt
import json
for n in (1000000, 10000000):
    for big in (64,128,256):
        try:
            d=json.loads(open(f"gpurun_out/r4z/bench_{n}_{big}.json").read().strip().splitlines()[-1])
            print(n, big, round(d["ms_per_step"],3), {k:round(x,3) for k,x in d["config"]["kernel_group_ms"].items()})
        except Exception as e: print(n, big, "ERR", e)
PY
