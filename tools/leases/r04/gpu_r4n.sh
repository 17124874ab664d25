#!/bin/bash
# round 4, lease n: k_group1 — records per thread in registers x probe sequences walked together
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4n; mkdir -p $O
cd $R
python -c "import bench; bench.gen_fasta(10000000,150,1)"
for v in ${VARIANTS:-p8b4 p8b8 p10b5 p12b6 p12b4 p12b4c}; do
  lib=$R/swarm_amd/lib/libswarm_amd_$v.so; [ $v = base ] && lib=$R/swarm_amd/lib/libswarm_amd.so
  echo "$v: $(SWARM_AMD_LIB=$lib timeout 200 python tools/experiments/time_build.py 2>$O/$v.err | tail -1)" | tee -a $O/variants.txt
done
