#!/bin/bash
# lease r6z15: configs[2] (-d 1 -f at 10 M) three times: was the 0.387 s of lease z14 the build or the box?
for i in 1 2 3; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs2 > /dev/null 2>&1
  python -c "
import json
d=json.load(open('bench_detail.json'))['config']['configs2']; print(d['pipeline_total_s'], d['pipeline_seconds'], d['fastidious_kernels_ms'])"
done
