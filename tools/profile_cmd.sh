#!/bin/bash
# tools/profile_cmd.sh <tag> <cmd...> — rocprofv3 counter passes for any command of the library (run on the GPU box through gpurun):
# one --pmc group per run, with --kernel-trace only (never with the API / memory-copy trace domains).  FETCH_SIZE and WRITE_SIZE in
# passes of their own, as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes.  Per-kernel averages -> gpurun_out/<tag>/pmc_kernels.json
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o k -- "$@" > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $group -d "$OUT/pmc$i" -o k -- "$@" > "$OUT/pmc$i.log" 2>&1
  find "$OUT/pmc$i" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc$i.csv" \;
done
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
python $REPO/tools/summarize_pmc_kernels.py "$OUT" "$OUT/pmc_kernels.json"
