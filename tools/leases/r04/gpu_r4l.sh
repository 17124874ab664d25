#!/bin/bash
# round 4, lease l: where k_group1's time goes — the kernel cut short after each of its stages (tools/experiments/time_build.py)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4l; mkdir -p $O
cd $R
python -c "import bench; bench.gen_fasta(10000000,150,1)"
for v in base stop1 stop2 stop3; do
  lib=$R/swarm_amd/lib/libswarm_amd_$v.so; [ $v = base ] && lib=$R/swarm_amd/lib/libswarm_amd.so
  echo "$v: $(SWARM_AMD_LIB=$lib timeout 200 python tools/experiments/time_build.py 2>$O/$v.err | tail -1)" | tee -a $O/stages.txt
  echo "$v nodup: $(SWA_D1_NO_DUP=1 SWARM_AMD_LIB=$lib timeout 200 python tools/experiments/time_build.py 2>$O/$v.nodup.err | tail -1)" | tee -a $O/stages.txt
done
