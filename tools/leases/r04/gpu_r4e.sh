#!/bin/bash
# round 4, lease e: per-kernel profile, group-cap sweep on the heavy-tailed set, 1 M step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
KSTATS_LINES=48 bash tools/kstats.sh r4e_step10M python $R/bench.py --steps 20 --warmup 3 --no-extras
cp $R/gpurun_out/r4e_step10M_kernel_stats.csv $O/ 2>/dev/null
cd $R
for cap in 4096 1024 256; do
  SWA_D1_GROUP_CAP=$cap timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail > $O/bench_cap$cap.json 2> $O/bench_cap$cap.err
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --extras none > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
for cap in (4096,1024,256):
    try:
        d=json.loads(open(f"gpurun_out/r4e/bench_cap{cap}.json").read().strip().splitlines()[-1])
        c=d["config"]["heavy_tail"]; print("cap",cap, round(c["ms_per_step"],3), {a:round(b,3) for a,b in c["kernel_group_ms"].items()})
    except Exception as e: print(cap,"ERR",e)
d=json.loads(open("gpurun_out/r4e/bench.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], {k:round(v["ms"],3) for k,v in d["roofline"]["kernels"].items()})
c=d["config"].get("configs1"); print("configs1", c and round(c["ms_per_step"],3), c and {a:round(b,3) for a,b in c["kernel_group_ms"].items()})
PY
