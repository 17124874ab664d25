#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
for n in 10000000 1000000; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --per-gpu $n > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
print($n, round(d["ms_per_step"],3), d["config"]["kernel_group_ms"])
PY
done
timeout 600 python tools/bench_extra.py 10000000 5 0 0.1 > $O/heavy.json 2> $O/heavy.err; cat $O/heavy.json | cut -c1-900
bash tools/kstats.sh r3k_heavy python tools/bench_extra.py 10000000 3 0 0.1
cut -d, -f1-4 $R/gpurun_out/r3k_heavy_kernel_stats.csv | cut -c1-120 | head -14
