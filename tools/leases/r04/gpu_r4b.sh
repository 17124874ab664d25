#!/bin/bash
# round 4, lease b: width classes (per-group record width), table route retired
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_lengths_gpu.py tests/test_guard_gpu.py -x -q -m gpu > $O/tests_lengths.log 2>&1; echo "tests_lengths rc=$?" >> $O/status.txt
tail -15 $O/tests_lengths.log
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_multi_gpu.py -x -q -m gpu > $O/tests_d1.log 2>&1; echo "tests_d1 rc=$?" >> $O/status.txt
tail -8 $O/tests_d1.log
timeout 200 python tools/stress/run.py 48 200000 > $O/stress.log 2>&1; echo "stress rc=$?" >> $O/status.txt
tail -12 $O/stress.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r4b/bench.json").read().strip().splitlines()[-1])
    print("headline", d["ms_per_step"], {k:round(v["ms"],3) for k,v in d["roofline"]["kernels"].items()})
    for k in ("configs1","skewed","heavy_tail","d1_x400","d1_x460","mixed_lengths"):
        c=d["config"].get(k)
        if isinstance(c,dict) and "ms_per_step" in c: print(k, round(c["ms_per_step"],3), c.get("anchor_width_nt"), c.get("ns_per_nucleotide"), {a:round(b,3) for a,b in c["kernel_group_ms"].items()})
        else: print(k, c)
    print("whole_run", d["config"].get("whole_run"))
except Exception as e: print("ERR", e)
PY
tail -5 $O/bench.err
cat $O/status.txt
