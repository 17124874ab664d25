#!/bin/bash
# lease r6r: one partition level less for the links (k_csr_rows: a workgroup per bucket of up to 2^14 sources): parity, then the step wide / narrow
O=gpurun_out/r6r; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt)
for mode in wide narrow wide narrow; do
  SWA_D1_CSR_ROWS=$mode python bench.py --no-extras --steps 20 --warmup 5 > $O/step_$mode.json 2> $O/step_$mode.err || tail -3 $O/step_$mode.err
  python -c "
import json; d=json.load(open('$O/step_$mode.json')); print('$mode', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
for mode in wide narrow; do
  SWA_D1_CSR_ROWS=$mode python bench.py --no-extras --steps 20 --warmup 5 --per-gpu 1000000 > $O/step1M_$mode.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/step1M_$mode.json')); print('1M $mode', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
