#!/bin/bash
# round 3, final: kernel stats + counters of the FINAL fastidious (configs[2]) and d >= 2 (configs[3]) kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3fin2
mkdir -p $O
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
  python $R/tools/summarize_kernels.py $O config$c $O/config${c}_kernels_pmc.json 0.01 > /dev/null 2>&1
  rm -f $O/config${c}_pmc*.csv
done
ls -la $O
head -8 $O/config2_kernel_stats.csv | cut -c1-150
head -8 $O/config3_kernel_stats.csv | cut -c1-150
