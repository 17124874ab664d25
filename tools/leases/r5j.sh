#!/bin/bash
# lease r5j — what decides between the short and the long process exit (experiment A / B); 280 cold runs on this box;
# kernel statistics of configs[3] (d = 3) and configs[2] (fastidious) before this round's work on them; skewed_70
O=gpurun_out/r5j; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
cat $FA > /dev/null
run() {
  local label=$1; shift
  echo "---- $label"
  for i in 1 2 3 4 5 6 7 8 9 10; do
    sleep 1; s=${EPOCHREALTIME/./}
    env "$@" SWARM_AMD_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "results written" | grep "^\[t" | tr '\n' ' '
    e=${EPOCHREALTIME/./}; echo " wall_ms $(( (e - s) / 1000 ))"
  done
}
{
run default X=1
run A_result_arrays_stay_mapped SWARM_AMD_EXPERIMENT_EXIT=A
run B_unmapped_then_3ms SWARM_AMD_EXPERIMENT_EXIT=B
} > $O/exit.txt 2>&1
bash tools/stress/cold_runs.sh 280 > $O/cold.txt 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras skewed_70,configs2,configs3 > $O/bench_extras.json 2> $O/bench_extras.err
KSTATS_LINES=30 timeout 600 bash tools/kstats.sh r5j_configs3 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/kstats_configs3.txt 2>&1
cp gpurun_out/r5j_configs3_kernel_stats.csv $O/ 2>/dev/null
cat $O/exit.txt; tail -n 1 $O/cold.txt | cut -c1-300; head -12 $O/kstats_configs3.txt | cut -c1-160
