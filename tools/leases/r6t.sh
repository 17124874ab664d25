#!/bin/bash
# lease r6t: kernel statistics and counters of configs[2] (10 M -f) and configs[3] (1 M x 400, d = 3) on the round's final build
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6t; mkdir -p $O
KSTATS_LINES=30 timeout 600 bash tools/kstats.sh r6t_c3 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/c3_head.txt 2>&1
KSTATS_LINES=30 timeout 600 bash tools/kstats.sh r6t_c2 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs2 > $O/c2_head.txt 2>&1
cp $R/gpurun_out/r6t_c2_kernel_stats.csv $R/gpurun_out/r6t_c3_kernel_stats.csv $O/ 2>/dev/null
timeout 900 bash tools/profile_cmd.sh r6t_c3k python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/c3_pmc.log 2>&1
timeout 900 bash tools/profile_cmd.sh r6t_c2k python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs2 > $O/c2_pmc.log 2>&1
cp $R/gpurun_out/r6t_c3k/pmc_kernels.json $O/c3_pmc_kernels.json 2>/dev/null; cp $R/gpurun_out/r6t_c2k/pmc_kernels.json $O/c2_pmc_kernels.json 2>/dev/null
head -14 $O/c3_head.txt | cut -c1-140; head -14 $O/c2_head.txt | cut -c1-140; ls -la $O
