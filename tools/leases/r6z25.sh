#!/bin/bash
# lease r6z25: the tiled kernel's items cut into parts of up to eight column tiles: parity (stream / d1 / fullsize tests), then the Zipf set and its kernel statistics
O=$PWD/gpurun_out/r6z25_out; mkdir -p $O
(timeout 1200 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
for rep in 1 2; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail,skewed_70 > /dev/null 2>$O/err.txt
  python - <<P
import json
d=json.load(open('bench_detail.json'))['config']
for k in ('heavy_tail','skewed_70'):
    v=d.get(k,{}); g=v.get('kernel_group_ms',{}); print(k, round(v.get('ms_per_step',0),3), 'pairs', round(g.get('pairs0',0),3), round(g.get('pairs1',0),3), v.get('neighbour_links'), v.get('error'))
P
done
KSTATS_LINES=6 timeout 600 bash tools/kstats.sh r6z25k python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras heavy_tail > /dev/null 2>&1
