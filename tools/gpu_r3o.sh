#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3o
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
run10() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_10M_$tag.json 2> $O/bench_10M_$tag.err
  python - $O/bench_10M_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "10M ms", round(d["ms_per_step"], 3), {k: round(v["ms"], 3) for k, v in d["roofline"]["kernels"].items()})
except Exception as e:
    print("bench parse failed", e)
PY
}
run10 large SWA_D1_GROUPS=large
run10 large_t4096 SWA_D1_GROUPS=large SWA_D1_PART_TILE=4096
run10 small SWA_D1_GROUPS=small
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
for mode, batch in (("large", "4"), ("large", "2"), ("large", "1"), ("small", "4")):
    os.environ["SWA_D1_GROUPS"] = mode
    os.environ["SWA_D1_PAIR_BATCH"] = batch
    a = argparse.Namespace(length=150, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, 1_000_000, 10)
    print(mode, "batch", batch, "1M", round(r["ms_per_step"], 4), {k: round(v, 3) for k, v in r["kernel_group_ms"].items()})
PY
tail -3 $O/x.err
