#!/bin/bash
# lease r5q — k_align_wfa with its history as a ring of the steps a step looks back to: the d >= 2 suites,
# configs[3] timing, kernel statistics
O=gpurun_out/r5q; mkdir -p $O; R=$PWD
timeout 1200 python -m pytest tests/test_dn_gpu.py tests/test_scan_gpu.py tests/test_cli_gpu.py tests/test_multi_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "d3 or x400" > $O/tests_fullsize.txt 2>&1
for mode in lds; do
  SWA_DN_PAIRS=$mode timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/bench_$mode.json 2> $O/bench_$mode.err
done
KSTATS_LINES=12 timeout 600 bash tools/kstats.sh r5q_configs3 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs1 --extras configs3 > $O/kstats_configs3.txt 2>&1
cp gpurun_out/r5q_configs3_kernel_stats.csv $O/ 2>/dev/null
for f in $O/tests.txt $O/tests_fullsize.txt; do grep -E "passed|failed|error" $f | tail -n 2; done
python - <<'PY'
import json
for m in ("lds",):
    l=[x for x in open(f"gpurun_out/r5q/bench_{m}.json") if x.startswith("{")]
    c=json.loads(l[-1])["config"]["configs3"]
    print(m, c.get("clustering_seconds"), c.get("gpu_kernels_ms"), c.get("qgram_comparisons"), c.get("aligned_pairs"), c.get("swarms"), c.get("error"))
PY
head -8 $O/kstats_configs3.txt | cut -c1-150
