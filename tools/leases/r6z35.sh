#!/bin/bash
# lease r6z35: the alignments' work list ordered by the pairs' q-gram distance (SWA_DN_ALIGN_ORDER=0/1): d >= 2 tests, configs[3] A/B, kernel stats
O=$PWD/gpurun_out/r6z35_out; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -n 3 -k "dn or d2 or d3 or graph or align or cli or fullsize or multi" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
for ao in 0 1 0 1; do
  SWA_DN_ALIGN_ORDER=$ao timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > /dev/null 2>&1
  python -c "
import json
d=json.load(open('bench_detail.json'))['config']['configs3']; print('order=$ao', d['clustering_seconds'], d['gpu_kernels_ms'], d['aligned_pairs'], d['swarms'])"
done
SWA_DN_ALIGN_ORDER=1 KSTATS_LINES=8 timeout 300 bash tools/kstats.sh r6z35k python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 2>&1 | grep "k_align\|k_dg_work\|radix\|onesweep" | awk -F, '{print $1,$2,$4}' | cut -c1-140
