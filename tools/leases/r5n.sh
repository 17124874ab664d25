#!/bin/bash
# lease r5n — configs[2] pipeline after the writer / sums / flags changes; the whole run, 10 runs; 280 cold runs on this box
O=gpurun_out/r5n; mkdir -p $O; R=$PWD
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs2 > $O/bench_configs2.json 2> $O/bench_configs2.err
timeout 600 python -m pytest tests/test_fastidious_gpu.py tests/test_cli_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
FA=/tmp/swa_bench_10000000x150_s1.fa
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
{
for i in 1 2 3 4 5 6 7 8 9 10; do sleep 1; s=${EPOCHREALTIME/./}; SWARM_AMD_TIMING=1 ./swarm_amd/bin/swarm -d 1 -o /tmp/o.txt -l /dev/null $FA 2>&1 | grep -E "results written|read and ordered|Clustering|Writing swarms" | grep "^\[t" | tr '\n' ' '; e=${EPOCHREALTIME/./}; echo " wall_ms $(( (e - s) / 1000 ))"; done
md5sum /tmp/o.txt
} > $O/runs.txt 2>&1
bash tools/stress/cold_runs.sh 280 > $O/cold.txt 2>&1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r5n/bench_configs2.json") if x.startswith("{")]
c=json.loads(l[-1])["config"]["configs2"]; print(c.get("pipeline_seconds"), c.get("pipeline_total_s"), c.get("counters_equal_reference_log"), c.get("error"))
PY
grep -E "passed|failed" $O/tests.txt | tail -n 1; cat $O/runs.txt | cut -c1-230; tail -n 1 $O/cold.txt | cut -c1-200
