#!/bin/bash
# lease r6z12: k_csr_bucket<8> held to 80 VGPRs (six workgroups a CU, 32 bytes of scratch)
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10M', round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"; done
KSTATS_LINES=3 bash tools/kstats.sh r6z12k python $PWD/bench.py --steps 20 --warmup 3 --no-extras 2>&1 | grep csr | awk -F, '{print $1,$2,$4,$6,$7}'
