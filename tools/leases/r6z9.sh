#!/bin/bash
# lease r6z9: the counter visits of the tiled / big-group paths as atomicInc with nothing computed behind them: the sets
# with large groups (heavy_tail: Zipf families; skewed_70, v4_like: window mode), then the stream tests
O=$PWD/gpurun_out/r6z9_out; mkdir -p $O
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail,skewed_70,v4_like,mixed_lengths,d1_x400 > $O/line.json 2>$O/err.txt; cp bench_detail.json $O/detail.json
python - <<P
import json
d=json.load(open('$O/detail.json'))['config']
for k in ('heavy_tail','skewed_70','v4_like','mixed_lengths','d1_x400'):
    v=d.get(k,{}); print(k, v.get('ms_per_step'), v.get('kernel_group_ms'), v.get('neighbour_links'), v.get('error'))
P
(timeout 1200 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10M', round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"; done
