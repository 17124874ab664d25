#!/bin/bash
# lease r6i: key records through the multi-GPU exchange: parity (routed tests, multi-rank tests, sharded bench on one GPU), then rank 0's share of 8 x 10 M
O=gpurun_out/r6i; mkdir -p $O
(timeout 1200 python -m pytest tests/test_d1_gpu.py tests/test_multi_gpu.py tests/test_bench_sharded_gpu.py tests/test_guard_gpu.py -x -q -n 3 > $O/tests.txt 2>&1; tail -5 $O/tests.txt)
for b in records routed records routed; do
  python bench.py --simulate-world 8 --build $b --steps 10 --warmup 3 --no-extras > $O/sim8_$b.json 2> $O/sim8_$b.err || tail -5 $O/sim8_$b.err
  python - <<PY
import json
d=json.load(open('$O/sim8_$b.json')); print('$b', d['ms_per_step'], d['roofline'].get('kernel_ms'))
PY
done
