#!/bin/bash
# round 3, call B: the streaming index build / CSR: triage against the oracle, then A/B timing and kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
rocm-smi --showserial 2>/dev/null | grep -i serial > $O/serial.txt; timeout 600 python tools/check_stream.py 200000 > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log
tail -30 $O/check.log
if grep -q "check rc=0" $O/check.log; then
  for v in "stream stream" "table table"; do
    set -- $v
    SWA_D1_BUILD=$1 SWA_D1_CSR=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
    python - <<PY
import json
d=json.loads(open("$O/bench_$1_$2.json").read().strip().splitlines()[-1])
print("$1 $2", round(d["ms_per_step"],3), d["config"]["phase_ms"], d["config"]["neighbour_links"])
PY
  done
  SWA_D1_BUILD=stream SWA_D1_CSR=stream timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --per-gpu 1000000 > $O/bench1M_stream.json 2>> $O/bench_stream_stream.err
  SWA_D1_BUILD=table SWA_D1_CSR=table timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --per-gpu 1000000 > $O/bench1M_table.json 2>> $O/bench_table_table.err
  python - <<PY
import json
for t in ("stream","table"):
    d=json.loads(open("$O/bench1M_%s.json"%t).read().strip().splitlines()[-1])
    print("1M",t, round(d["ms_per_step"],3), d["config"]["phase_ms"])
PY
  bash tools/kstats.sh r3b_stream python $R/bench.py --steps 4 --warmup 1 --no-extras
  cp $R/gpurun_out/r3b_stream_kernel_stats.csv $O/
  bash tools/kstats.sh r3b_stream1M python $R/bench.py --steps 4 --warmup 1 --no-extras --per-gpu 1000000
  cp $R/gpurun_out/r3b_stream1M_kernel_stats.csv $O/
  ( time timeout 1500 python -m pytest tests/ -x -q -m gpu -k "not 100m" ) > $O/tests.log 2>&1
  tail -8 $O/tests.log
fi
