#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
rocm-smi --showserial 2>/dev/null | grep -i serial > $O/serial.txt
( time timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_multi_gpu.py tests/test_fullsize_gpu.py -x -q -k "not 100m" ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -3
for n in 10000000 1000000; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --per-gpu $n > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
print($n, round(d["ms_per_step"],3), d["config"]["phase_ms"], d["config"]["kernel_group_ms"])
PY
done
bash tools/kstats.sh r3i_10M python $R/bench.py --steps 4 --warmup 1 --no-extras
cp $R/gpurun_out/r3i_10M_kernel_stats.csv $O/
