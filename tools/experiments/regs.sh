#!/bin/bash
# tools/experiments/regs.sh KERNEL_SUBSTRING [flags...] : VGPRs / scratch bytes / LDS of the kernels of d1.hip whose name contains the substring
cd "$(dirname "$0")/../../swarm_amd/csrc"
k=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -x hip --cuda-device-only -S -o /tmp/regs_$$.s d1.hip 2>/dev/null
awk -v k="$k" '/^[ \t]*\.amdhsa_kernel/ {name=$2; on = index(name, k) > 0} on && /amdhsa_private_segment_fixed_size|amdhsa_next_free_vgpr|amdhsa_group_segment_fixed_size/ {print name, $1, $2}' /tmp/regs_$$.s
rm -f /tmp/regs_$$.s
