#!/bin/bash
# round 4, lease w: partition workgroups of 256 / 512 / 1024 threads; k_group1 with the cheaper scan tail — bench + parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4w; mkdir -p $O
cd $R
for t in 256 512 1024; do
  SWA_D1_PART_THREADS=$t timeout 300 python bench.py --steps 10 --warmup 2 --no-extras > $O/bench_t$t.json 2> $O/bench_t$t.err
done
python - <<'PY' | tee $O/summary.txt
import json
for t in (256,512,1024):
    try:
        d=json.loads(open(f"gpurun_out/r4w/bench_t{t}.json").read().strip().splitlines()[-1])
        print(t, round(d["ms_per_step"],3), {k:round(x["ms"],3) for k,x in d["roofline"]["kernels"].items()}, d["roofline"]["kernel"][:30], round(d["roofline"]["frac"],3))
    except Exception as e: print(t, "ERR", e)
PY
timeout 1500 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_guard_gpu.py tests/test_lengths_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
for t in 512 1024; do
  SWA_D1_PART_THREADS=$t timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_lengths_gpu.py -m gpu -x -q > $O/tests_t$t.log 2>&1; echo "tests t$t rc=$?" | tee -a $O/summary.txt
done
tail -3 $O/tests.log
