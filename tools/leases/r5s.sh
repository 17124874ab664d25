#!/bin/bash
# lease r5s — the round's closing artefacts on the final build: 12 whole runs (the exit's two modes), the driver's bench line,
# kernel statistics of the step at 10 M, the PMC passes, smoke(), the whole GPU suite
O=gpurun_out/r5s; mkdir -p $O; R=$PWD
FA=/tmp/swa_bench_10000000x150_s1.fa
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
cat $FA > /dev/null
{
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 1; s=${EPOCHREALTIME/./}; SWARM_AMD_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "results written|read and ordered|context created|Clustering|Writing swarms" | grep "^\[t" | tr '\n' ' '; e=${EPOCHREALTIME/./}; echo " wall_ms $(( (e - s) / 1000 ))"; done
md5sum /tmp/o.txt
echo "---- reference -t 16"; s=${EPOCHREALTIME/./}; oracle/_ref/swarm -d 1 -t 16 -o /tmp/ro.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; md5sum /tmp/ro.txt
} > $O/whole_run.txt 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" > $O/status.txt
KSTATS_LINES=40 timeout 600 bash tools/kstats.sh r5s_step10M python $R/bench.py --steps 20 --warmup 3 --no-extras > $O/kstats.txt 2>&1; cp gpurun_out/r5s_step10M_kernel_stats.csv $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/status.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests_all rc=$?" >> $O/status.txt
cat $O/status.txt; tail -n 2 $O/tests_all.log; cat $O/whole_run.txt | cut -c1-250; tail -c 400 $O/bench_default.json
