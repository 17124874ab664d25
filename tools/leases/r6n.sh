#!/bin/bash
# lease r6n: identical sequences met by the pair kernels' prefix pass (no fingerprints, no second table in k_group1): the whole GPU suite, then the step
O=gpurun_out/r6n; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt)
python bench.py --no-extras --steps 20 --warmup 5 > $O/bench_step.json 2> $O/bench_step.err || tail -5 $O/bench_step.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6n/bench_step.json')); print(d['ms_per_step'], d['roofline'].get('kernel_ms'))
PY
python bench.py --no-extras --steps 10 --warmup 3 --per-gpu 1000000 > $O/bench_step_1M.json 2> $O/bench_step_1M.err; python -c "
import json; d=json.load(open('gpurun_out/r6n/bench_step_1M.json')); print('1M', d['ms_per_step'], d['roofline'].get('kernel_ms'))"
python bench.py --simulate-world 8 --steps 10 --warmup 3 --no-extras > $O/sim8_records.json 2> $O/sim8.err; python -c "
import json; d=json.load(open('gpurun_out/r6n/sim8_records.json')); print('sim8', d['ms_per_step'], d['roofline'].get('kernel_ms'))"
