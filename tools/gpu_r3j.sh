#!/bin/bash
# round 3, call J: the whole GPU suite, the full bench line, kernel stats of the step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
rocm-smi --showserial 2>/dev/null | grep -i serial > $O/serial.txt
( time timeout 2000 python -m pytest tests/ -x -q -m gpu ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err ); tail -c 600 $O/bench_full.json
bash tools/kstats.sh r3j_10M python $R/bench.py --steps 4 --warmup 1 --no-extras
bash tools/kstats.sh r3j_1M python $R/bench.py --steps 4 --warmup 1 --no-extras --per-gpu 1000000
cp $R/gpurun_out/r3j_*_kernel_stats.csv $O/
