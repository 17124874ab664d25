#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3z
mkdir -p $O
cd $R
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
for n in (10_000_000, 1_000_000):
  for rb in ("0", "3", "4", "5", "6"):
    os.environ["SWA_D1_XCD_RUN_BITS"] = rb
    a = argparse.Namespace(length=150, seed=1)
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, n, 10)
    print("run_bits", rb, n, round(r["ms_per_step"], 4), {k: round(v, 3) for k, v in r["kernel_group_ms"].items() if "partition" in k}, r["neighbour_links"], flush=True)
PY
tail -3 $O/x.err
