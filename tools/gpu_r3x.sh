#!/bin/bash
# final evidence run of round 3: the whole GPU suite, the full bench line, kernel traces of the 10 M and 1 M steps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3final2
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -q -m gpu ) > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
( time timeout 900 python bench.py ) > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.json; tail -3 $O/bench_full.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k10 -o k -- python $R/bench.py --steps 5 --warmup 2 --no-extras > $O/k10.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1 -o k -- python $R/bench.py --steps 5 --warmup 2 --no-extras --per-gpu 1000000 > $O/k1.log 2>&1
find $O/k10 $O/k1 -name "*.db" -delete 2>/dev/null
find $O/k10 $O/k1 -name "*kernel_trace.csv" -delete 2>/dev/null
ls $O/k10 $O/k1
