#!/bin/bash
# lease r5h — the whole run with the scratch block and the word pools released beside the GPU's work; then the whole GPU
# suite on this build and the driver's bench line (all extras: roofline by binding resource, first_step, whole_run, configs2/3)
O=gpurun_out/r5h; mkdir -p $O
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
cat $FA > /dev/null
{
for i in 1 2 3 4; do
  echo "---- default run $i"; sleep 1; s=${EPOCHREALTIME/./}
  SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "^\["
  e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"
done
echo "---- quiet, 8 runs"
for i in 1 2 3 4 5 6 7 8; do sleep 1; s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
md5sum /tmp/o.txt
echo "---- with -w and -s (words kept), 2 runs"
for i in 1 2; do sleep 1; s=${EPOCHREALTIME/./}; ./swarm_amd/bin/swarm -d 1 -o /tmp/o2.txt -s /tmp/s2.txt -w /tmp/w2.txt -l /dev/null $FA; e=${EPOCHREALTIME/./}; echo "wall_ms $(( (e - s) / 1000 ))"; done
md5sum /tmp/o2.txt /tmp/s2.txt
} > $O/runs.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x -k "not 100" > $O/tests.txt 2>&1
timeout 1200 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
grep -E "wall_ms|^----" $O/runs.txt | tr '\n' ' '; echo; grep -E "passed|failed" $O/tests.txt | tail -n 2; tail -c 600 $O/bench_default.err
