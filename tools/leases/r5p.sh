#!/bin/bash
# lease r5p — 280 cold runs on a fourth box; configs[2] again
O=gpurun_out/r5p; mkdir -p $O; R=$PWD
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs2 > $O/bench_configs2.json 2> $O/bench_configs2.err
FA=/tmp/swa_bench_10000000x150_s1.fa
python -c "import bench; print(bench.gen_fasta(10000000,150,1))" > $O/gen.txt 2>&1
bash tools/stress/cold_runs.sh 280 > $O/cold.txt 2>&1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r5p/bench_configs2.json") if x.startswith("{")]
c=json.loads(l[-1])["config"]["configs2"]; print(c.get("pipeline_seconds"), c.get("pipeline_total_s"), c.get("counters_equal_reference_log"), c.get("error"))
PY
tail -n 1 $O/cold.txt | cut -c1-200
