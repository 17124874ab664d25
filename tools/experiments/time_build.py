#!/usr/bin/env python3
"""tools/experiments/time_build.py — the index-build half of the d=1 step alone (keys, key partition, groups), kernel group
times from the library's HIP events.  For A/B of k_group1 variants (SWARM_AMD_LIB=...), including ablated ones whose
group lists are wrong on purpose: no network call follows, so nothing reads them.

    python tools/experiments/time_build.py [--n 10000000] [--length 150] [--steps 10]
"""
import argparse, json, sys, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench

def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--length", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from swarm_amd import Context, HostDb
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    bench.gen_fasta(args.n, args.length, args.seed)
    hdb = HostDb(Path(tempfile.gettempdir()) / f"swa_bench_{args.n}x{args.length}_s{args.seed}.fa")
    to_dev = lambda a, t: torch.from_numpy(np.ascontiguousarray(a).view(t)).to(dev)
    t_seqs = to_dev(np.concatenate([hdb.seqs, np.zeros(2, dtype=np.uint64)]), np.int64)
    t_off, t_len, t_ab = to_dev(hdb.seq_off, np.int64), to_dev(hdb.seqlen, np.int32), to_dev(hdb.abundance, np.int64)
    ctx = Context(0, torch.cuda.current_stream(dev).cuda_stream)
    ctx.attach_db(t_seqs, t_off, t_len, t_ab, hdb.longest)
    ctx.timing_enable(True)
    rows = []
    for i in range(args.steps + 2):
        ctx.d1_index_build(0, args.n)
        st = ctx.timing_read_stream()
        if i >= 2:
            rows.append([st[0], st[1], st[2]])
    m = np.mean(np.array(rows), axis=0)
    print(json.dumps({"n": args.n, "length": args.length, "keys_ms": round(float(m[0]), 4), "partition_keys_ms": round(float(m[1]), 4),
                      "groups_ms": round(float(m[2]), 4)}))

if __name__ == "__main__":
    main()
