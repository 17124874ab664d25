#!/bin/bash
# round 4, lease d: per-kernel profile of the step after the launch trims
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4d; mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --extras configs1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4d/bench.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], {k:round(v["ms"],3) for k,v in d["roofline"]["kernels"].items()})
c=d["config"].get("configs1"); print("configs1", c and round(c["ms_per_step"],3), c and {a:round(b,3) for a,b in c["kernel_group_ms"].items()})
PY
KSTATS_LINES=45 bash tools/kstats.sh r4d_step10M python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-configs1
cp $R/gpurun_out/r4d_step10M_kernel_stats.csv $O/ 2>/dev/null
timeout 300 python -m pytest tests/test_stream_gpu.py tests/test_lengths_gpu.py tests/test_guard_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/status.txt
tail -4 $O/tests.log
cat $O/status.txt
