#!/bin/bash
# lease r6l: the whole GPU suite and the default bench line on the build of the afternoon (agglomeration, streamed reader, key records, host diet)
O=gpurun_out/r6l; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt)
python bench.py > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; cp bench_detail.json $O/
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6l/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))
for k in ('cpu_baseline','cpu_baseline_10M','whole_run','first_step_ms','host_seam_ms','configs1','configs2','configs3'): print(k, d.get(k))
PY
