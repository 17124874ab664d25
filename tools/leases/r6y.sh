#!/bin/bash
# lease r6y: the wavefront alignment kernel with 16-lane groups (four pairs a wave) where the band fits: parity, then configs[3] with 16 / 32 lanes
O=gpurun_out/r6y; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt)
for lanes in 16 32 16 32; do
  SWA_ALIGN_WFA_LANES=$lanes python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/c3_$lanes.json 2>/dev/null; cp bench_detail.json $O/c3_detail_$lanes.json
  python -c "
import json
d=json.load(open('$O/c3_detail_$lanes.json'))['config']['configs3']; print('$lanes', d['clustering_seconds'], d['gpu_kernels_ms'], d['aligned_pairs'], d['swarms'])"
done
SWA_ALIGN_WFA_LANES=16 KSTATS_LINES=6 bash tools/kstats.sh r6y_c3 python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 | cut -c1-130
