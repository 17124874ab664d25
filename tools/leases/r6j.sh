#!/bin/bash
# lease r6j: where the -f run (configs[2]) and the d = 3 run (configs[3]) spend their time outside the kernels
O=gpurun_out/r6j; mkdir -p $O
python -c "
import bench
print(bench.gen_fasta(10000000,150,1,1,0.3)); print(bench.gen_fasta(1000000,400,1,3,0.0))" > $O/gen.txt 2>&1
F2=$(sed -n 1p $O/gen.txt); F3=$(sed -n 2p $O/gen.txt)
{
for i in 1 2 3; do
  echo "== -f run $i"
  s=${EPOCHREALTIME/./}
  SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -f -o /tmp/of.txt -l /dev/null $F2 2>&1 | grep -E "^\[t |^\[cluster\]"
  e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; sleep 1
done
} > $O/config2_timeline.txt 2>&1
{
for i in 1 2 3; do
  echo "== d3 run $i"
  s=${EPOCHREALTIME/./}
  SWARM_AMD_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 3 -o /tmp/o3.txt -l /dev/null $F3 2>&1 | grep -E "^\[t |^\[cluster|^\[dn"
  e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; sleep 1
done
} > $O/config3_timeline.txt 2>&1
sed -n 1,60p $O/config2_timeline.txt; sed -n 1,60p $O/config3_timeline.txt
