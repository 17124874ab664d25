#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
for mode in large small; do
  SWA_D1_GROUPS=$mode timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_10M_$mode.json 2> $O/bench_10M_$mode.err
  python - $O/bench_10M_$mode.json $mode <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "10M ms", d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items()})
except Exception as e:
    print("bench parse failed", e)
PY
done
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
for mode in ("large", "small"):
    os.environ["SWA_D1_GROUPS"] = mode
    a = argparse.Namespace(length=150, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, 1_000_000, 10)
    print(mode, "1M", json.dumps({"ms": r["ms_per_step"], "groups": r["kernel_group_ms"], "links": r["neighbour_links"]}))
PY
tail -3 $O/x.err
