#!/bin/bash
# round 3, call A: ceilings (ubench + PMC calibration) and BASELINE configs[4] through one context / 8 ranks on device 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
nproc > $O/host.txt; free -g >> $O/host.txt; df -h /tmp >> $O/host.txt; rocm-smi --showmeminfo vram >> $O/host.txt 2>&1
timeout 300 tools/ubench_lines $O/ubench.json 2> $O/ubench.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc_$c -o u -- $R/tools/ubench_lines > $O/pmc_$c.log 2>&1
  find $O/pmc_$c -name '*counter_collection.csv' -exec cp {} $O/ubench_pmc_$c.csv \;
  rm -rf $O/pmc_$c
done
cd $R
( time timeout 1500 python -m pytest tests/test_fullsize_gpu.py -k 100m -x -q ) > $O/test100m.log 2>&1
tail -5 $O/test100m.log
