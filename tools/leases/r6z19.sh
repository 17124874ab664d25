#!/bin/bash
# lease r6z19: the tiled kernel with its next column tile's lines and the ids of the one after fetched ahead: A/B against the build before on the Zipf set
for rep in 1 2; do for v in _head ""; do
  SWARM_AMD_LIB=$PWD/swarm_amd/lib/libswarm_amd$v.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras heavy_tail,skewed_70 > /dev/null 2>&1
  python - <<P
import json
d=json.load(open('bench_detail.json'))['config']
for k in ('heavy_tail','skewed_70'):
    v=d.get(k,{}); g=v.get('kernel_group_ms',{}); print('lib$v', k, round(v.get('ms_per_step',0),3), 'pairs', round(g.get('pairs0',0),3), round(g.get('pairs1',0),3), v.get('neighbour_links'), v.get('error'))
P
done; done
for v in _head ""; do SWARM_AMD_LIB=$PWD/swarm_amd/lib/libswarm_amd$v.so KSTATS_LINES=6 bash tools/kstats.sh r6z19k$v python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras heavy_tail > /dev/null 2>&1; done
