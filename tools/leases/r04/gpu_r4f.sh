#!/bin/bash
# round 4, lease f: batched probing in k_group1, one status copy per phase; dup-search cost; tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_lengths_gpu.py tests/test_guard_gpu.py -q -m gpu -x > $O/tests_d1.log 2>&1; echo "tests_d1 rc=$?" >> $O/status.txt
tail -4 $O/tests_d1.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --extras none > $O/bench.json 2> $O/bench.err
SWA_D1_NO_DUP=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs1 --extras none > $O/bench_nodup.json 2> $O/bench_nodup.err
python - <<'PY'
import json
for f in ("bench","bench_nodup"):
    d=json.loads(open(f"gpurun_out/r4f/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], {k:round(v["ms"],3) for k,v in d["roofline"]["kernels"].items()})
    c=d["config"].get("configs1")
    if c: print("configs1", round(c["ms_per_step"],3), {a:round(b,3) for a,b in c["kernel_group_ms"].items()})
PY
KSTATS_LINES=16 bash tools/kstats.sh r4f_step10M python $R/bench.py --steps 20 --warmup 3 --no-extras
cat $O/status.txt
