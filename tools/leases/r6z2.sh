#!/bin/bash
# lease r6z2: where the pair kernels' time goes — the same step with bundles that skip their turns (x1), their line fetches (x2), both (x3)
# (libraries built with -DSWA_PAIR_EXPERIMENT=n beside the real one; profiling only)
O=gpurun_out/r6z2; mkdir -p $O
for x in 0 1 2 3; do
  L=$PWD/swarm_amd/lib/libswarm_amd_x$x.so; [ $x = 0 ] && L=$PWD/swarm_amd/lib/libswarm_amd.so
  SWARM_AMD_LIB=$L KSTATS_LINES=3 bash tools/kstats.sh r6z2_x$x python $PWD/bench.py --steps 10 --warmup 2 --no-extras 2>&1 | grep "group_pairs" | cut -c1-140
done
