#!/bin/bash
O=gpurun_out/r6f; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
{
for mode in default 0 default; do
for i in 1 2 3; do
  echo "== $mode"
  env $( [ $mode = default ] && echo X=1 || echo SWARM_AMD_PIN_RESULTS=$mode ) SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\[cluster|pinned|Clustering|Building|uploaded"
done
done
echo "== default, OMP_NUM_THREADS=1 SWARM_AMD_HOST_THREADS=4"
OMP_NUM_THREADS=1 SWARM_AMD_HOST_THREADS=4 SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\[cluster|pinned|Clustering|Building|uploaded"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
} > $O/cluster_laps.txt 2>&1
cat $O/cluster_laps.txt
