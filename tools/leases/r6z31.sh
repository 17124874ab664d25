#!/bin/bash
# lease r6z31: pivot marks in the tiled kernel now that its items are even: parity (stream / d1 / fullsize), A/B by SWA_D1_PIVOT_MARKS on the Zipf and V4-like sets, kernel stats
O=$PWD/gpurun_out/r6z31_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
for pm in 0 1 0 1; do
  SWA_D1_PIVOT_MARKS=$pm timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail,v4_like > /dev/null 2>$O/err$pm.txt
  python - <<P
import json
d=json.load(open('bench_detail.json'))['config']
for k in ('heavy_tail','v4_like'):
    v=d.get(k,{}); g=v.get('kernel_group_ms',{}); print('marks=$pm', k, round(v.get('ms_per_step',0),3), 'pairs', round(g.get('pairs0',0),3), round(g.get('pairs1',0),3), v.get('neighbour_links'), v.get('error'))
P
done
for pm in 0 1; do SWA_D1_PIVOT_MARKS=$pm KSTATS_LINES=6 timeout 400 bash tools/kstats.sh r6z31k$pm python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras heavy_tail > /dev/null 2>&1; done
