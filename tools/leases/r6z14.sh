#!/bin/bash
# lease r6z14: the default bench line of the evening build with its side file kept; rank 0's share of an 8 x 10 M job
O=$PWD/gpurun_out/r6z14_out; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/err.txt; cp bench_detail.json $O/bench_detail.json
tail -c 300 $O/bench_default.json; echo
for b in records; do timeout 600 python bench.py --simulate-world 8 --build $b --steps 10 --warmup 3 --no-extras > $O/sim8_$b.json 2>> $O/err.txt; python -c "
import json
d=json.load(open('$O/sim8_$b.json')); print('$b', d['ms_per_step'], d['roofline'].get('kernel_ms'))"; done
