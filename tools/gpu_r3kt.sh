#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3kt
mkdir -p $O
cd $R
python - <<'PY' 2> $O/x.err
import sys, json, argparse, os
sys.path.insert(0, ".")
import bench, torch
torch.cuda.set_device(0)
a = argparse.Namespace(length=150, seed=1)
for tile in ("4096", "2048"):
    os.environ["SWA_D1_KEY_TILE"] = tile
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, 10_000_000, 10)
    print("key tile", tile, round(r["ms_per_step"], 3), {k: round(v, 3) for k, v in r["kernel_group_ms"].items()}, r["neighbour_links"], flush=True)
del os.environ["SWA_D1_KEY_TILE"]
for cap in ("4096", "2048", "1024"):
    os.environ["SWA_D1_GROUP_CAP"] = cap
    r = bench.extra_measurement(torch, torch.device("cuda", 0), 0, a, 10_000_000, 5, zipf=0.1)
    print("heavy_tail cap", cap, round(r["ms_per_step"], 3), {k: round(v, 3) for k, v in r["kernel_group_ms"].items() if k.startswith("pairs") or k.startswith("plain")}, r["neighbour_links"], flush=True)
PY
tail -3 $O/x.err
