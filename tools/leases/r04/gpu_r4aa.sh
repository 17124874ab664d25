#!/bin/bash
# round 4, lease aa: where the key partition's scatter spends its time (the kernel cut short after each stage)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4aa; mkdir -p $O
cd $R
python -c "import bench; bench.gen_fasta(10000000,150,1)"
for t in 512 256; do for v in ps0 ps1 ps2 ps3; do
  echo "$v threads=$t: $(SWA_D1_PART_THREADS=$t SWARM_AMD_LIB=$R/swarm_amd/lib/libswarm_amd_$v.so timeout 200 python tools/experiments/time_build.py 2>$O/$v.$t.err | tail -1)" | tee -a $O/stages.txt
done; done
