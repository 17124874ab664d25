#!/bin/bash
# lease r6z3: pair kernels three bundles deep (lines of i + 1, ids of i + 2, item of i + 3 on their way while i is compared): 4 workgroups a CU (104 VGPRs) / 5 (96 + 40 bytes of scratch)
O=$PWD/gpurun_out/r6z3; mkdir -p $O
for v in "" _w4; do
  L=$PWD/swarm_amd/lib/libswarm_amd$v.so
  SWARM_AMD_LIB=$L KSTATS_LINES=3 bash tools/kstats.sh r6z3$v python $PWD/bench.py --steps 10 --warmup 2 --no-extras 2>&1 | grep "group_pairs" | cut -c1-140
  SWARM_AMD_LIB=$L python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('neighbour_links'))"
done
(timeout 1700 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt)
