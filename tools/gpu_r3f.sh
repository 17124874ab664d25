#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
rocm-smi --showserial 2>/dev/null | grep -i serial > $O/serial.txt
for i in 1 2 3 4 5 6; do
  timeout 300 python tools/check_stream.py 200000 stream > $O/check_$i.log 2>&1; echo "run $i rc=$?" | tee -a $O/summary.txt
  grep -E "lines:|DIFFERENT|missing" $O/check_$i.log | head -8 | tee -a $O/summary.txt
done
