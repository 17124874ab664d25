#!/bin/bash
# lease r6z10: which kernel of the conserved-flank set (skewed_70: window mode, 200 nt) slowed down — kernel stats of that measurement alone
KSTATS_LINES=30 bash tools/kstats.sh r6z10k python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras skewed_70 2>&1 | awk -F, '{print $1,$2,$4,$6,$7}' | grep "8, 1>\|Name" | cut -c1-200
