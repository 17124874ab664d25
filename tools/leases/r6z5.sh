#!/bin/bash
# lease r6z5: pair kernels — links emitted with one branch a turn and without the link counts nobody reads; three bundles deep; parity + step
O=$PWD/gpurun_out/r6z5; mkdir -p $O
KSTATS_LINES=3 bash tools/kstats.sh r6z5 python $PWD/bench.py --steps 10 --warmup 2 --no-extras 2>&1 | grep "group_pairs" | cut -c1-140
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('neighbour_links'))"; done
cd $PWD; (timeout 1700 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt)
