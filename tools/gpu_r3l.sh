#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
for cap in 4096 2048 1024 512; do
  for mt in 1 0; do
    SWA_D1_GROUP_CAP=$cap SWA_D1_MEMBER_TABLE=$mt timeout 600 python tools/bench_extra.py 10000000 4 0 0.1 > $O/heavy_${cap}_$mt.json 2> $O/heavy_${cap}_$mt.err
    python - <<PY
import json
d=json.loads(open("$O/heavy_${cap}_$mt.json").read().strip().splitlines()[-1])
k=d["kernel_group_ms"]
print("cap $cap member_table $mt", round(d["ms_per_step"],2), "pairs", round(k["pairs0"]+k["pairs1"],2), "network", round(d["network_kernels_ms"],2), "hash+table", round(k["plain_kernel_and_table"],2), "groups", round(k["groups"],2), d["neighbour_links"])
PY
  done
done
