#!/bin/bash
# round 4, lease t: partition output loop unrolled, CSR rows sorted in registers — bench + parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4t; mkdir -p $O
cd $R
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench.json 2> $O/bench.err
python - <<'PY' | tee $O/summary.txt
import json
d=json.loads(open("gpurun_out/r4t/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["ms_per_step"],3), {k:round(x["ms"],3) for k,x in d["roofline"]["kernels"].items()})
PY
timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_guard_gpu.py tests/test_lengths_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log
