#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_d1_gpu.py tests/test_stream_gpu.py tests/test_cli_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
python - <<'PY' > $O/x400.json 2> $O/x400.err
import sys, json, argparse
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
for L, n in ((400, 1_000_000), (250, 1_000_000), (150, 1_000_000)):
    a = argparse.Namespace(length=L, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, n, 5)
    print(json.dumps({"L": L, "ms": r["ms_per_step"], "groups": r["kernel_group_ms"], "links": r["neighbour_links"]}))
PY
cat $O/x400.json; tail -3 $O/x400.err
SWA_D1_BUILD=table SWA_D1_CSR=table python - <<'PY' 2>/dev/null
import sys, json, argparse
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
a = argparse.Namespace(length=400, seed=1)
r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, 1_000_000, 5)
print("table route 400:", json.dumps({"ms": r["ms_per_step"], "net": r["network_kernels_ms"], "links": r["neighbour_links"]}))
PY
