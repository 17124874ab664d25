#!/bin/bash
# lease r6z33: column tiles an item of the tiled kernel covers: 4 / 8 / 16 (libraries built -DSWA_TILED_COLS=n) on the Zipf set
for rep in 1 2; do for v in _t4 "" _t16; do
  SWARM_AMD_LIB=$PWD/swarm_amd/lib/libswarm_amd$v.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail > /dev/null 2>&1
  python - <<P
import json
d=json.load(open('bench_detail.json'))['config']
v=d.get('heavy_tail',{}); g=v.get('kernel_group_ms',{}); print('lib$v', round(v.get('ms_per_step',0),3), 'pairs', round(g.get('pairs0',0),3), round(g.get('pairs1',0),3), v.get('neighbour_links'), v.get('error'))
P
done; done
