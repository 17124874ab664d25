#!/bin/bash
# lease r5k — short sequences served from the member table (skewed_70, the d1 / lengths suites); kernel statistics of
# configs[3] (d = 3) and configs[2] (fastidious) before this round's work on them
O=gpurun_out/r5k; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_d1_gpu.py tests/test_lengths_gpu.py tests/test_stream_gpu.py tests/test_guard_gpu.py tests/test_cli_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras skewed_70,skewed,heavy_tail > $O/bench_extras.json 2> $O/bench_extras.err
KSTATS_LINES=40 timeout 600 bash tools/kstats.sh r5k_configs3 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/kstats_configs3.txt 2>&1
cp gpurun_out/r5k_configs3_kernel_stats.csv $O/ 2>/dev/null
KSTATS_LINES=40 timeout 600 bash tools/kstats.sh r5k_configs2 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs2 > $O/kstats_configs2.txt 2>&1
cp gpurun_out/r5k_configs2_kernel_stats.csv $O/ 2>/dev/null
grep -E "passed|failed" $O/tests.txt | tail -n 2; head -30 $O/kstats_configs3.txt | cut -c1-150
