#!/bin/bash
# tools/scale_check.sh [n=10000000] [gpus=8] — first thing to run on a multi-GPU node.
#
# The d=1 path on N GPUs from the drop-in command line (SWARM_AMD_DEVICES=0,..,N-1: swa_multi_*, one rank + host
# thread per GPU inside the process, routed index build with grouped ncclSend / ncclRecv, link lists gathered with
# RCCL over xGMI) against the SAME run on one GPU: -o and -j must be byte-identical, the N-GPU run must really have
# used RCCL (swa_multi_uses_rccl: the log line "multi: N ranks, exchange = rccl"), and the per-phase times of both
# runs are printed (SWARM_AMD_TIMING=1).  Exit status 0 = everything matched.
set -u
N=${1:-10000000}
G=${2:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${TMPDIR:-/tmp}/scale_check.$$
mkdir -p "$OUT"
FA=$(cd "$R" && python3 -c "import bench; print(bench.gen_fasta($N, 150, 1))") || exit 2
DEV=$(seq -s, 0 $((G-1)))
echo "== one GPU"
SWARM_AMD_TIMING=1 "$R/swarm_amd/bin/swarm" -d 1 -o "$OUT/one.o" -j "$OUT/one.j" -l "$OUT/one.log" "$FA" 2> "$OUT/one.err" || { cat "$OUT/one.err"; exit 1; }
grep -E '^\[t' "$OUT/one.err"
for build in routed streamed; do
  echo "== $G GPUs ($DEV), index build $build"
  SWARM_AMD_MULTI_BUILD=$build SWARM_AMD_MULTI_REPORT=1 SWARM_AMD_TIMING=1 SWARM_AMD_DEVICES=$DEV "$R/swarm_amd/bin/swarm" -d 1 \
      -o "$OUT/$build.o" -j "$OUT/$build.j" -l "$OUT/$build.log" "$FA" 2> "$OUT/$build.err" || { cat "$OUT/$build.err"; exit 1; }
  grep -E '^\[t|^multi:' "$OUT/$build.err"
  grep -q "exchange = rccl" "$OUT/$build.err" || { echo "FAIL: the $G-GPU run did not use RCCL"; exit 1; }
  cmp "$OUT/one.o" "$OUT/$build.o" || { echo "FAIL: -o differs ($build)"; exit 1; }
  cmp "$OUT/one.j" "$OUT/$build.j" || { echo "FAIL: -j differs ($build)"; exit 1; }
done
echo "OK: $G-GPU runs (routed and streamed index build, RCCL exchange) byte-identical to one GPU on $N amplicons"
rm -rf "$OUT"
