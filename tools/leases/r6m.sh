#!/bin/bash
# lease r6m: does releasing the pinned order array before the exit decide who takes the address space apart?  30 runs each, interleaved
O=gpurun_out/r6m; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
{
for i in $(seq 1 30); do
  for m in none 1 2; do
    if [ $m = none ]; then unset SWARM_AMD_EXIT_EXPERIMENT; else export SWARM_AMD_EXIT_EXPERIMENT=$m; fi
    s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}
    echo "$m $(( (e - s) / 1000 ))"; sleep 0.6
  done
done
} > $O/exit_experiment.txt 2>&1
for m in none 1 2; do echo "$m: $(grep "^$m " $O/exit_experiment.txt | awk '{print $2}' | sort -n | tr '\n' ' ')"; done
