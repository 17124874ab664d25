#!/bin/bash
# round 4, lease c: after the k_keys / list fixes — the d1 suites that lease b did not reach, quick bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py tests/test_lengths_gpu.py tests/test_guard_gpu.py -q -m gpu > $O/tests_d1.log 2>&1; echo "tests_d1 rc=$?" >> $O/status.txt
tail -8 $O/tests_d1.log
timeout 600 python -m pytest tests/test_multi_gpu.py tests/test_fullsize_gpu.py tests/test_fastidious_gpu.py tests/test_cli_gpu.py tests/test_ref_gpu.py -q -m gpu -k "not 100" > $O/tests_more.log 2>&1; echo "tests_more rc=$?" >> $O/status.txt
tail -8 $O/tests_more.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --extras heavy_tail,mixed_lengths,d1_x460 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r4c/bench.json").read().strip().splitlines()[-1])
    print("headline", d["ms_per_step"], {k:round(v["ms"],3) for k,v in d["roofline"]["kernels"].items()})
    for k in ("configs1","heavy_tail","d1_x460","mixed_lengths"):
        c=d["config"].get(k)
        if isinstance(c,dict) and "ms_per_step" in c: print(k, round(c["ms_per_step"],3), c.get("anchor_width_nt"), {a:round(b,3) for a,b in c["kernel_group_ms"].items()})
        else: print(k, c)
except Exception as e: print("ERR", e)
PY
cat $O/status.txt
