#!/bin/bash
# round 3, call G: optimised group / CSR kernels, multi-GPU d >= 2 tests, the new bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
rocm-smi --showserial 2>/dev/null | grep -i serial > $O/serial.txt
timeout 600 python tools/check_stream.py 200000 > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log
grep -E "DIFFERENT|rc=|lines:" $O/check.log | head
timeout 300 python tools/check_index.py 200000 > $O/index.log 2>&1; echo "index rc=$?" >> $O/index.log; tail -4 $O/index.log
( time timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_stream_gpu.py -x -q ) > $O/tests_multi.log 2>&1; tail -5 $O/tests_multi.log
for n in 10000000 1000000; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --per-gpu $n > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
print($n, round(d["ms_per_step"],3), d["config"]["kernel_group_ms"], d["roofline"]["step"])
PY
done
bash tools/kstats.sh r3g_10M python $R/bench.py --steps 4 --warmup 1 --no-extras
bash tools/kstats.sh r3g_1M python $R/bench.py --steps 4 --warmup 1 --no-extras --per-gpu 1000000
cp $R/gpurun_out/r3g_*_kernel_stats.csv $O/
( time timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err ); tail -c 3000 $O/bench_full.json
