#!/bin/bash
# tools/experiments/variant.sh NAME "-DX=1 ..." : libswarm_amd_NAME.so = this tree with d1.hip compiled under extra flags
# (A/B of kernel variants on one lease: SWARM_AMD_LIB=swarm_amd/lib/libswarm_amd_NAME.so python bench.py ...)
set -e
cd "$(dirname "$0")/../../swarm_amd/csrc"
name=$1; shift
make -s -j 16 all
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter "$@" -x hip -c -o build/variant_d1_$name.o d1.hip
objs=$(ls build/*.o | grep -v "^build/d1.o$" | grep -v "^build/variant_" | grep -v "^build/asan_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libswarm_amd_$name.so $objs build/variant_d1_$name.o \
  -L$(dirname $(g++ -print-file-name=libgomp.so)) -lgomp -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../lib/libswarm_amd_$name.so
