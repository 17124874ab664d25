#!/bin/bash
# round 4, lease u: duplicate table only for records with company — timing, bench, parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
python -c "import bench; bench.gen_fasta(10000000,150,1)"
echo "base: $(timeout 200 python tools/experiments/time_build.py 2>$O/base.err | tail -1)" | tee -a $O/summary.txt
echo "nodup: $(SWA_D1_NO_DUP=1 timeout 200 python tools/experiments/time_build.py 2>$O/nodup.err | tail -1)" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench.json 2> $O/bench.err
python - <<'PY' | tee -a $O/summary.txt
import json
d=json.loads(open("gpurun_out/r4u/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["ms_per_step"],3), {k:round(x["ms"],3) for k,x in d["roofline"]["kernels"].items()})
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_guard_gpu.py tests/test_lengths_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log
