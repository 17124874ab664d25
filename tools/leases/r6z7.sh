#!/bin/bash
# lease r6z7: the first level of the link partition with tiles over the segments laid end to end (SWA_D1_LINK_SPAN=0: chunk by chunk, as before);
# pair kernels three bundles deep with the new emission.  Parity suite, then A/B of the step
O=$PWD/gpurun_out/r6z7_out; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt)
for rep in 1 2; do for sp in 0 1; do
  SWA_D1_LINK_SPAN=$sp python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('span=$sp', round(d['ms_per_step'],4), d['roofline']['kernel_ms'], d['config'].get('neighbour_links'))"
done; done
python bench.py --per-gpu 1000000 --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1M', round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"
KSTATS_LINES=14 bash tools/kstats.sh r6z7k python $PWD/bench.py --steps 20 --warmup 3 --no-extras 2>&1 | awk -F, '{print $1,$2,$4,$6,$7}' | cut -c1-170
