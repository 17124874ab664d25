#!/bin/bash
# lease r6z1: pair kernels — chains without the dwords that cannot change the answer (PASS 0: the lowest 2 NW end-aligned dwords,
# PASS 1: the forward dwords beyond the first window), class sizes in registers, item + ids of the NEXT bundle fetched ahead
O=gpurun_out/r6z1; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -x -q -n 3 > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt)
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-extras > $O/step10M_$i.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/step10M_$i.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('neighbour_links'))"
done
python bench.py --per-gpu 1000000 --steps 30 --warmup 5 --no-extras > $O/step1M.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/step1M.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
KSTATS_LINES=12 bash tools/kstats.sh r6z1_step10M python $PWD/bench.py --steps 20 --warmup 3 --no-extras | cut -c1-150
