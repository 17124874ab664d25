#!/bin/bash
# lease r6z27: k_dg_pairs_lds asks for a stage's queries a stage ahead: d >= 2 tests, configs[3] twice, kernel stats
O=$PWD/gpurun_out/r6z27_out; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q -n 3 -k "dn or d2 or d3 or graph or align or cli or fullsize or multi" > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
for i in 1 2; do
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/c3.json 2>/dev/null
  python -c "
import json
d=json.load(open('bench_detail.json'))['config']['configs3']; print(d['clustering_seconds'], d['gpu_kernels_ms'], d['aligned_pairs'], d['swarms'], d['qgram_comparisons'])"
done
KSTATS_LINES=8 timeout 400 bash tools/kstats.sh r6z27k python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 2>&1 | grep "k_dg_pairs\|k_align" | awk -F, '{print $1,$2,$4}' | cut -c1-150
