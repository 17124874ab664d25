#!/bin/bash
# tools/profile_d1.sh — rocprofv3 passes for the d=1 bench (run on the GPU box through gpurun).
#   pass 1: --kernel-trace --stats           (per-kernel time)
#   pass 2..: --pmc, one counter group per run (HBM bytes, L2 hit rate, SQ occupancy/issue)
# Outputs land under gpurun_out/<tag>/ ; copy the summaries you keep into profiles/.
set -u
TAG=${1:-prof}
STEPS=${2:-3}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps $STEPS --warmup 1 --no-extras"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o d1 -- $BENCH > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;

i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $group -d "$OUT/pmc$i" -o d1 -- $BENCH > "$OUT/pmc$i.log" 2>&1
  find "$OUT/pmc$i" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc$i.csv" \;
done
# keep the merged directory small: drop everything but the CSV summaries and logs
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la "$OUT"
python $REPO/tools/summarize_pmc.py "$OUT" $((STEPS+1)) "${3:-10000000x150_s1}" "$OUT/d1_step_pmc.json"
head -8 "$OUT/kernel_stats.csv"
