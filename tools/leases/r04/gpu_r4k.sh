#!/bin/bash
# round 4, lease k: d >= 2 after the host-side changes, label propagation with pointer jumping, whole-run timeline
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
SWARM_AMD_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --extras configs3 > $O/bench_c3.json 2> $O/bench_c3.err
grep "\[dn" $O/bench_c3.err | head -20
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4k/bench_c3.json").read().strip().splitlines()[-1])
c=d["config"].get("configs3"); print({k:c[k] for k in ("clustering_seconds","gpu_kernels_ms","aligned_pairs")} if isinstance(c,dict) and "clustering_seconds" in c else c)
PY
FA=$(ls /tmp/swa_bench_10000000x150_s1.fa)
for i in 1 2 3; do
  ( time SWARM_AMD_TIMING=1 SWARM_AMD_DB_TIMING=1 SWARM_AMD_CLUSTER_TIMING=1 swarm_amd/bin/swarm -d 1 -o /tmp/out_$i.txt $FA ) > $O/whole_run_$i.log 2>&1
done
cat $O/whole_run_2.log | tail -45
md5sum /tmp/out_1.txt /tmp/out_2.txt
timeout 900 python -m pytest tests/test_dn_gpu.py tests/test_cli_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "not 100" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/status.txt
tail -4 $O/tests.log
cat $O/status.txt
