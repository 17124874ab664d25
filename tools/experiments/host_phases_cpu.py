#!/usr/bin/env python3
"""tools/experiments/host_phases_cpu.py [n] [--fastidious] — the HOST phases behind the GPU (clustering tables, light flags,
graft, writers) timed on this machine's cores, without a GPU: the network and the graft candidates come from the C oracle
(oracle/, through tests/support.py — a development aid like the tests, nothing the product path touches) and are cached under
the temp dir.  What it is for: changes to host/cluster_d1.cpp and host/out.h can be measured where there is no GPU, and
their output compared with the unmodified reference's (`oracle/_ref/swarm`) at a size where the parallel paths run."""
import hashlib
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench            # noqa: E402
import support as S     # noqa: E402
from swarm_amd import D1Clusters, HostDb   # noqa: E402


def timed(label, fn, reps=1):
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{label:28s} {best * 1e3:9.1f} ms" + (f"  (best of {reps})" if reps > 1 else ""), flush=True)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2_000_000
    fast = "--fastidious" in sys.argv
    fa = bench.gen_fasta(n, 150, 1, 1, 0.3 if fast else 0.0)
    tmp = Path(tempfile.gettempdir())
    tag = f"{n}_{'f' if fast else 'p'}"
    hdb = timed("HostDb", lambda: HostDb(fa))
    db = S.Db(headers=[], seqs=hdb.seqs, seq_off=hdb.seq_off, seqlen=hdb.seqlen, abundance=hdb.abundance, longest=int(hdb.seqlen.max()))
    net = tmp / f"swa_hostbench_net_{tag}.npz"
    if net.exists():
        z = np.load(net); off, nb = z["off"], z["nb"]
    else:
        off, nb, _ = timed("oracle network", lambda: S.oracle_d1_network(db))
        np.savez(net, off=off, nb=nb)
    cl = timed("host clustering", lambda: D1Clusters(hdb, off, nb))
    out = tmp / f"swa_hostbench_{tag}.out"
    args = ["-d", "1"]
    if fast:
        flags, stats = timed("light_flags", cl.light_flags)
        gf = tmp / f"swa_hostbench_graft_{tag}.npy"
        if gf.exists():
            graft = np.load(gf)
        else:
            graft, _ = timed("oracle fastidious", lambda: S.oracle_fastidious(db, flags))
            np.save(gf, graft)
        print("grafts", timed("graft", lambda: cl.graft(graft)))
        args.append("-f")
    timed("write_swarms", lambda: cl.write_swarms(out), reps=5)
    md5 = hashlib.md5(out.read_bytes()).hexdigest()
    ref_md5 = tmp / f"swa_hostbench_ref_{tag}.md5"
    if not ref_md5.exists() and S.have_reference():
        ref_out = tmp / f"swa_hostbench_ref_{tag}.out"
        timed("reference", lambda: subprocess.run([str(S.ref_swarm_bin())] + args + ["-t", str(os.cpu_count() or 1), "-o", str(ref_out), "-l", "/dev/null", str(fa)], check=True))
        ref_md5.write_text(hashlib.md5(ref_out.read_bytes()).hexdigest())
        ref_out.unlink()
    print("output md5", md5, "reference", ref_md5.read_text() if ref_md5.exists() else "n/a")


if __name__ == "__main__":
    main()
