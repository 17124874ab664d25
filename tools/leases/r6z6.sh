#!/bin/bash
# lease r6z6: A/B on one box — the committed pair kernels (one bundle ahead: libswarm_amd_head.so) against three bundles deep + the new emission
for rep in 1 2; do for v in _head ""; do
  L=$PWD/swarm_amd/lib/libswarm_amd$v.so
  SWARM_AMD_LIB=$L python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$v', round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"
done; done
for v in _head ""; do
  SWARM_AMD_LIB=$PWD/swarm_amd/lib/libswarm_amd$v.so KSTATS_LINES=9 bash tools/kstats.sh r6z6k$v python $PWD/bench.py --steps 20 --warmup 3 --no-extras 2>&1 | grep "group_pairs\|k_part_scatter<0, 4096\|k_part_hist" | awk -F, '{print $1,$4}' | cut -c1-150
done
