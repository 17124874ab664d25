#!/bin/bash
# lease r6z29: the dynamic-LDS attribute of the wide pair kernels set once per context: the 400 / 460-nt sets three times, the length tests
for i in 1 2 3; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras d1_x400,d1_x460 > /dev/null 2>&1
  python -c "
import json
d=json.load(open('bench_detail.json'))['config']
print({k: round(d[k]['ms_per_step'],3) for k in ('d1_x400','d1_x460')})"
done
(timeout 900 python -m pytest tests/test_d1_gpu.py tests/test_lengths_gpu.py tests/test_stream_gpu.py -m gpu -x -q -n 3 2>&1 | tail -2)
