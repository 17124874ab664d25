#!/bin/bash
# round 4, lease s: k_group1 — tail records asked for early, members staged in LDS, no loop-invariant spills
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
python -c "import bench; bench.gen_fasta(10000000,150,1)"
for v in base stop1 stop2 stop3 stop4; do
  lib=$R/swarm_amd/lib/libswarm_amd_$v.so; [ $v = base ] && lib=$R/swarm_amd/lib/libswarm_amd.so
  echo "$v: $(SWARM_AMD_LIB=$lib timeout 200 python tools/experiments/time_build.py 2>$O/$v.err | tail -1)" | tee -a $O/stages.txt
done
echo "base nodup: $(SWA_D1_NO_DUP=1 timeout 200 python tools/experiments/time_build.py 2>$O/nodup.err | tail -1)" | tee -a $O/stages.txt
timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_guard_gpu.py tests/test_lengths_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/stages.txt
tail -5 $O/tests.log
