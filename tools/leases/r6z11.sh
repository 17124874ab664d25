#!/bin/bash
# lease r6z11: k_csr_bucket without the 32-wide register sort (113 -> 89 VGPRs: five workgroups a CU instead of four), grid = resident workgroups
O=$PWD/gpurun_out/r6z11_out; mkdir -p $O
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10M', round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail > $O/line.json 2>$O/err.txt; python - <<P
import json
d=json.load(open('bench_detail.json'))['config']
for k in ('heavy_tail',):
    v=d.get(k,{}); print(k, v.get('ms_per_step'), v.get('kernel_group_ms'), v.get('error'))
P
KSTATS_LINES=3 bash tools/kstats.sh r6z11k python $PWD/bench.py --steps 20 --warmup 3 --no-extras 2>&1 | grep csr | awk -F, '{print $1,$2,$4,$6,$7}'
(timeout 1500 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
