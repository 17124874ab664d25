#!/bin/bash
# lease r6z17: the tiled pair kernel decides pairs by their members' marks against the group's pivot (SWA_D1_PIVOT_MARKS=0/1 A/B on the Zipf set), parity first
O=$PWD/gpurun_out/r6z17_out; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
for pm in 0 1; do
  SWA_D1_PIVOT_MARKS=$pm python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail,v4_like > /dev/null 2>$O/err$pm.txt
  python - <<P
import json
d=json.load(open('bench_detail.json'))['config']
for k in ('heavy_tail','v4_like'):
    v=d.get(k,{}); g=v.get('kernel_group_ms',{}); print('marks=$pm', k, v.get('ms_per_step'), 'pairs', g.get('pairs0'), g.get('pairs1'), v.get('neighbour_links'), v.get('error'))
P
done
SWA_D1_PIVOT_MARKS=1 KSTATS_LINES=6 bash tools/kstats.sh r6z17k python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras heavy_tail 2>&1 | grep "tiled\|group_pairs" | awk -F, '{print $1,$2,$4}' | cut -c1-160
