#!/bin/bash
# lease r6z8: the partition's tile -> chunk table written with 32-bit arithmetic; A/B of the spanning tiles; kernel stats
for rep in 1 2; do for sp in 0 1; do
  SWA_D1_LINK_SPAN=$sp python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('span=$sp', round(d['ms_per_step'],4), d['roofline']['kernel_ms'], d['config'].get('neighbour_links'))"
done; done
python bench.py --per-gpu 1000000 --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1M', round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"
python bench.py --per-gpu 100000 --steps 30 --warmup 5 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('100k', round(d['ms_per_step'],4), d['roofline']['kernel_ms'])"
KSTATS_LINES=14 bash tools/kstats.sh r6z8k python $PWD/bench.py --steps 20 --warmup 3 --no-extras > /dev/null 2>&1
