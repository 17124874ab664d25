#!/bin/bash
# round 3, call H: tile / trip-count tweaks, heavy-tail + multi tests, kernel stats + PMC of the fastidious and d >= 2 kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
rocm-smi --showserial 2>/dev/null | grep -i serial > $O/serial.txt
( time timeout 1200 python -m pytest tests/test_stream_gpu.py tests/test_multi_gpu.py tests/test_d1_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -3
for n in 10000000 1000000; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --per-gpu $n > $O/bench_$n.json 2> $O/bench_$n.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
print($n, round(d["ms_per_step"],3), d["config"]["kernel_group_ms"])
PY
done
bash tools/kstats.sh r3h_10M python $R/bench.py --steps 4 --warmup 1 --no-extras
cp $R/gpurun_out/r3h_10M_kernel_stats.csv $O/
# configs[2] (10 M -f) and configs[3] (1 M x 400, d=3): kernel stats and counters of their own kernels
cat > /tmp/cfg.py <<'PY'
import sys, json
sys.path.insert(0, sys.argv[2])
import bench, argparse
a = argparse.Namespace(length=150, seed=1, per_gpu=10_000_000)
print(json.dumps(bench.config2_fastidious(a, 10_000_000) if sys.argv[1] == "2" else bench.config3_dn(a, 1_000_000, 400, 3)))
PY
cd /tmp && export TMPDIR=/tmp
for c in 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$c -o k -- python /tmp/cfg.py $c $R > $O/cfg$c.log 2>&1
  find $O/k$c -name '*kernel_stats.csv' -exec cp {} $O/config${c}_kernel_stats.csv \;
  rm -rf $O/k$c
  i=0
  for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $group -d $O/p$c$i -o p -- python /tmp/cfg.py $c $R > /dev/null 2>&1
    find $O/p$c$i -name '*counter_collection.csv' -exec cp {} $O/config${c}_pmc$i.csv \;
    rm -rf $O/p$c$i
  done
done
cd $R
head -12 $O/config2_kernel_stats.csv | cut -c1-160
head -12 $O/config3_kernel_stats.csv | cut -c1-160
