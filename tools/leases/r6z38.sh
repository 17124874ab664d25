#!/bin/bash
# lease r6z38: the alignments' order key with the length difference weighed in (SWA_DN_ALIGN_LEN_WEIGHT = 0 / 5 / 10 / 20)
for w in 0 5 10 20 0 10; do
  SWA_DN_ALIGN_LEN_WEIGHT=$w timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > /dev/null 2>&1
  python -c "
import json
d=json.load(open('bench_detail.json'))['config']['configs3']; print('weight=$w', d['clustering_seconds'], d['gpu_kernels_ms'], d['swarms'])"
done
