#!/bin/bash
# lease r6e: why the download into the pinned result arrays takes 9 ms in the command line and 0.75 ms in the experiment
O=gpurun_out/r6e; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
{
for mode in default touch 0; do
for i in 1 2 3; do
  echo "== $mode"
  env $( [ $mode = default ] && echo X=1 || echo SWARM_AMD_PIN_RESULTS=$mode ) SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\[cluster|pinned|Clustering|Building|uploaded"
done
done
} > $O/cluster_laps.txt 2>&1
cat $O/cluster_laps.txt
