#!/usr/bin/env python3
"""tools/stress/run.py [runs=200] [n=200000] — the hunt for the lost-links anomaly.

Starts tools/stress/first_step `runs` times — every one a fresh process, a fresh HIP runtime, the first launch of every
kernel of the d=1 step — under the runtime conditions only a first run has, in turn:
    warm-up of the context on a helper thread beside the FASTA read (as the command line does) / none
    AMD_SERIALIZE_KERNEL=3 (every kernel waits for the one before and is waited for) / unset
    64-nt anchor windows (what this set gets) / 32-nt windows (every 8th run: other groups, the same network)
and compares every stage's checksum (amplicon lines, member lists, group sizes, CSR) with the first run's and the CSR
with the C oracle's network.  Writes gpurun_out/stress/summary.json; exit 1 if any run differed or failed."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as S  # noqa: E402

M64 = (1 << 64) - 1


def mix(x):
    x ^= x >> 33; x = (x * 0xff51afd7ed558ccd) & M64; x ^= x >> 33; x = (x * 0xc4ceb9fe1a85ec53) & M64; x ^= x >> 33
    return x


def chain(words, tail=b""):
    h = 0x9E3779B97F4A7C15
    for w in words:
        h = mix(h ^ int(w))
    for b in tail:
        h = mix(h ^ b)
    return h


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    out_dir = os.path.join(ROOT, "gpurun_out", "stress")
    os.makedirs(out_dir, exist_ok=True)
    fa = f"/tmp/stress_{n}.fa"
    S.gen_fasta(fa, n, 150, 5)
    db = S.db_from_fasta(fa)
    woff, wnb, _ = S.oracle_d1_network(db)
    wnb = wnb.copy()
    for i in range(db.n):
        wnb[int(woff[i]):int(woff[i + 1])].sort()
    nb_bytes = wnb.astype(np.uint32).tobytes()
    whole = len(nb_bytes) // 8 * 8
    want = mix(chain(woff.astype(np.uint64)) ^ chain(np.frombuffer(nb_bytes[:whole], dtype=np.uint64), nb_bytes[whole:]))
    exe = os.path.join(ROOT, "tools", "stress", "first_step")
    results, bad = [], []
    reference = {}
    t0 = time.time()
    for r in range(runs):
        warm, serial, narrow = r & 1, (r >> 1) & 1, (r % 8) == 7
        env = dict(os.environ, STRESS_WARMUP=str(warm))
        env.pop("AMD_SERIALIZE_KERNEL", None)
        if serial:
            env["AMD_SERIALIZE_KERNEL"] = "3"
        env.pop("SWA_D1_ANCHOR_W", None)
        if narrow:
            env["SWA_D1_ANCHOR_W"] = "32"
        p = subprocess.run([exe, fa, f"{want:016x}"], capture_output=True, text=True, env=env, timeout=300)
        line = p.stdout.strip().splitlines()[0] if p.stdout.strip() else ""
        fields = dict(kv.split("=", 1) for kv in line.split() if "=" in kv)
        cond = f"warm{warm}_serial{serial}_{'w32' if narrow else 'auto'}"
        rec = {"run": r, "cond": cond, "rc": p.returncode, "line": line}
        ok = p.returncode == 0
        key = "w32" if narrow else "auto"
        if ok:
            ref = reference.setdefault(key, fields)
            differing = [k for k in fields if fields[k] != ref.get(k)]
            if differing:
                ok = False
                rec["differs_from_first_run_in"] = differing
        if not ok:
            rec["stdout"] = p.stdout[-2000:]
            rec["stderr"] = p.stderr[-2000:]
            bad.append(rec)
        results.append(rec)
    by_cond = {}
    for rec in results:
        c = by_cond.setdefault(rec["cond"], {"runs": 0, "bad": 0})
        c["runs"] += 1
        c["bad"] += 0 if rec["rc"] == 0 and "differs_from_first_run_in" not in rec else 1
    serial_txt = subprocess.run("rocm-smi --showserial 2>/dev/null | grep -i serial | head -2", shell=True, capture_output=True, text=True).stdout.strip()
    summary = {"runs": runs, "n": n, "oracle_csr": f"{want:016x}", "seconds": round(time.time() - t0, 1), "bad_runs": len(bad), "by_condition": by_cond,
               "reference_lines": {k: " ".join(f"{a}={b}" for a, b in v.items()) for k, v in reference.items()}, "bad": bad[:20], "gpu": serial_txt}
    with open(os.path.join(out_dir, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: summary[k] for k in ("runs", "seconds", "bad_runs", "by_condition", "reference_lines")}, indent=1))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
