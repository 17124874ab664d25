#!/bin/bash
# lease r6w: result pages faulted in by the library's workers before large downloads: the host seam (B1 with host buffers), configs[2]; parity
O=gpurun_out/r6w; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt)
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs1 --extras host_seam_ms,configs2,configs3 > $O/bench.json 2> $O/bench.err; cp bench_detail.json $O/
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6w/bench.json')); print(d['ms_per_step'], d.get('host_seam_ms')); print(d.get('configs2')); print(d.get('configs3'))
dd=json.load(open('gpurun_out/r6w/bench_detail.json')); print(dd['config']['configs2']['pipeline_seconds'])
PY
