#!/bin/bash
# tools/stress/cold_runs.sh RUNS [N=10000000] — the anomaly hunt at full size: RUNS fresh processes
# of the command line on the N x 150 set, each the first d = 1 step of a fresh HIP runtime on freshly allocated memory,
# under rotating runtime conditions (SDMA off, kernel arguments in device memory, one hardware queue, no code-object
# warm-up, poisoned HBM); every -o file must be the reference's (md5 from tests/golden/fullsize.json), and no run may
# have needed the guard's repeated step (its line on stderr).  Appends one JSON line to gpurun_out/stress/cold_runs.jsonl.
RUNS=${1:-100}; N=${2:-10000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/stress
FA=$(python -c "import bench; print(bench.gen_fasta($N,150,1))")
WANT=$(python -c "import json; print(json.load(open('tests/golden/fullsize.json'))['$N']['runs']['d1']['files']['o']['md5'])")
CONDS=("X=1" "HSA_ENABLE_SDMA=0" "HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=1" "SWARM_AMD_NO_WARMUP=1" "SWARM_AMD_POISON_MB=8192" "SWARM_AMD_NO_WARMUP=1 SWARM_AMD_POISON_MB=8192" "AMD_SERIALIZE_KERNEL=3")
declare -A ok bad retried
t0=$SECONDS
for ((r = 0; r < RUNS; ++r)); do
  c=${CONDS[$((r % ${#CONDS[@]}))]}
  env $c ./swarm_amd/bin/swarm -d 1 -o /tmp/cold_o.txt -l /dev/null "$FA" 2> /tmp/cold_err.txt
  rc=$?
  got=$(md5sum < /tmp/cold_o.txt | cut -d' ' -f1)
  if [ $rc -eq 0 ] && [ "$got" == "$WANT" ]; then ok[$c]=$((${ok[$c]:-0} + 1)); else bad[$c]=$((${bad[$c]:-0} + 1)); cp /tmp/cold_err.txt gpurun_out/stress/cold_failure_$r.txt; fi
  if grep -q "guard" /tmp/cold_err.txt; then retried[$c]=$((${retried[$c]:-0} + 1)); cp /tmp/cold_err.txt gpurun_out/stress/cold_guard_$r.txt; fi
done
serial=$(rocm-smi --showserial 2>/dev/null | grep -i "GPU\[0\]" | head -1 | awk "{print \$NF}")
{
printf '{"gpu_serial": "%s", "n": %s, "runs": %s, "seconds": %s, "want_md5": "%s", "by_condition": {' "$serial" "$N" "$RUNS" "$((SECONDS - t0))" "$WANT"
first=1
for c in "${CONDS[@]}"; do
  [ $first -eq 1 ] || printf ', '; first=0
  printf '"%s": {"identical": %s, "different_or_failed": %s, "guard_retries": %s}' "$c" "${ok[$c]:-0}" "${bad[$c]:-0}" "${retried[$c]:-0}"
done
printf '}}\n'
} | tee -a gpurun_out/stress/cold_runs.jsonl
