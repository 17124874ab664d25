#!/bin/bash
# lease r6z4: the pair kernels' comparisons without the emission of links (library built with -DSWA_PAIR_EXPERIMENT=4; profiling only)
for v in "" _x4; do
  L=$PWD/swarm_amd/lib/libswarm_amd$v.so
  SWARM_AMD_LIB=$L KSTATS_LINES=3 bash tools/kstats.sh r6z4$v python $PWD/bench.py --steps 10 --warmup 2 --no-extras 2>&1 | grep "group_pairs" | cut -c1-140
done
