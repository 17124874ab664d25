#!/bin/bash
# round 4, lease y: a bucket beyond 16-bit record numbers
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4y; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_lengths_gpu.py tests/test_stream_gpu.py tests/test_guard_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
tail -25 $O/tests.log
