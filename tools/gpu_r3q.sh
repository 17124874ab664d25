#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3q
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_d1_gpu.py -x -q ) > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -3
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
variants = [(3, 1, 4), (3, 64, 4), (6, 64, 4), (6, 64, 2), (6, 64, 1), (6, 1024, 2), (8, 64, 1), (8, 64, 2)]
for n in (10_000_000, 1_000_000):
  for bits, stride, batch in variants:
    os.environ["SWA_D1_PAIR_BATCH"] = str(batch)
    os.environ["SWA_D1_PAIR_SHARD_BITS"] = str(bits)
    os.environ["SWA_D1_SCHED_STRIDE"] = str(stride)
    a = argparse.Namespace(length=150, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, n, 8)
    print(n, "bits", bits, "stride", stride, "batch", batch, round(r["ms_per_step"], 4), {k: round(v, 3) for k, v in r["kernel_group_ms"].items() if k.startswith("pairs")}, r["neighbour_links"], flush=True)
PY
tail -3 $O/x.err
