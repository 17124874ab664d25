#!/usr/bin/env python3
"""Full-size golden data: runs the UNMODIFIED reference (oracle/_ref/swarm) on the BENCH sets
(bench.gen_fasta, deterministic: 1 M x 150 and 10 M x 150 for `-d 1`; the same sizes with 30 %
light amplicons — BASELINE configs[2] — for `-d 1 -f`) and keeps only a few hundred bytes per
run in tests/golden/fullsize.json:

  * md5 + size of each input FASTA (so a GPU box can tell that it regenerated the same set),
  * md5 + size of every output file (-o, -s, -j for `-d 1`; -o, -s, -i for `-d 1 -f`),
  * the deterministic summary lines of the log (swarm counts, fastidious counters:
    src/algod1.cc:1436-1438, 1469-1472, 1484-1487).

The -j file lists every link of the d=1 network (source, target by header, in db order,
src/algod1.cc:755-788), so its md5 pins the COMPLETE network, not a sample of rows.

Runs only where /root/reference was compiled (this container); minutes of CPU at 10 M.

    python tests/golden/make_fullsize.py [1000000 10000000] [d1 d1_f]
"""
from __future__ import annotations

import hashlib
import json
import re
import subprocess
import sys
import tempfile
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE.parent))
import bench  # noqa: E402
import support as S  # noqa: E402

KEEP = re.compile(r"^(Database info|Number of swarms|Largest swarm|Max generations|Heavy swarms|Light swarms|"
                  r"Total length|Bloom filter|Generated|Heavy variants|Got|Made|Results before)")
# run -> (swarm arguments, files kept, generator's light fraction, reference threads).  The fastidious
# runs use ONE thread: the reference clears Bloom bits with a plain `&=` from all its threads
# (src/bloomflex.cc:61-64), an update can be lost, and then a candidate is missed — "Got N graft
# candidates" (and, rarely, a graft) depends on thread timing.  -t 1 is the specification.
RUNS = {"d1": (["-d", "1"], "osj", 0.0, 8), "d1_f": (["-d", "1", "-f"], "osi", 0.3, 1)}
# BASELINE configs[3]: 1 M x 400, d = 3 (generator with up to 3 edits per amplicon); `d3` is only run for 1 M
RUNS_DN = {"d3": (["-d", "3"], "osi", 3, 8)}
FLAG = {"o": "-o", "s": "-s", "i": "-i", "j": "-j"}


def md5_of(path: Path) -> dict:
    h = hashlib.md5()
    with open(path, "rb") as fh:
        while True:
            chunk = fh.read(64 << 20)
            if not chunk:
                break
            h.update(chunk)
    return {"md5": h.hexdigest(), "bytes": path.stat().st_size}


def main() -> None:
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1_000_000, 10_000_000]
    only = [a for a in sys.argv[1:] if a in RUNS] or ([] if any(a in RUNS_DN for a in sys.argv[1:]) else list(RUNS))
    out_path = HERE / "fullsize.json"
    data = json.loads(out_path.read_text()) if out_path.exists() else {}
    for n in sizes:
        rec = data.get(str(n), {"runs": {}})
        rec.pop("threads", None)
        for name, (args, keep, light, threads) in RUNS.items():
            if name not in only:
                continue
            fasta = bench.gen_fasta(n, 150, 1, 1, light)
            with tempfile.TemporaryDirectory() as tmp:
                cmd = list(args) + ["-t", str(threads)]
                for k in keep:
                    cmd += [FLAG[k], f"{tmp}/{k}"]
                cmd += ["-l", f"{tmp}/log", str(fasta)]
                t0 = time.perf_counter()
                r = S.run_ref_swarm(cmd)
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr
                log = [ln for ln in Path(f"{tmp}/log").read_text().splitlines() if KEEP.match(ln)]
                rec["runs"][name] = {"args": args, "generator": f"bench.gen_fasta({n}, 150, 1, 1, {light})",
                                     "fasta": md5_of(fasta), "reference_threads": threads, "reference_seconds_here": round(dt, 2),
                                     "files": {k: md5_of(Path(f"{tmp}/{k}")) for k in keep}, "log": log}
                print(n, name, f"{dt:.1f} s", flush=True)
        for name, (args, keep, edits, threads) in RUNS_DN.items():
            if name not in sys.argv[1:] or n != 1_000_000:
                continue
            fasta = bench.gen_fasta(n, 400, 1, edits, 0.0)
            with tempfile.TemporaryDirectory() as tmp:
                cmd = list(args) + ["-t", str(threads)]
                for k in keep:
                    cmd += [FLAG[k], f"{tmp}/{k}"]
                cmd += ["-l", f"{tmp}/log", str(fasta)]
                t0 = time.perf_counter()
                r = S.run_ref_swarm(cmd)
                dt = time.perf_counter() - t0
                assert r.returncode == 0, r.stderr
                log = [ln for ln in Path(f"{tmp}/log").read_text().splitlines() if KEEP.match(ln)]
                rec["runs"][name] = {"args": args, "generator": f"bench.gen_fasta({n}, 400, 1, {edits}, 0.0)",
                                     "fasta": md5_of(fasta), "reference_threads": threads, "reference_seconds_here": round(dt, 2),
                                     "files": {k: md5_of(Path(f"{tmp}/{k}")) for k in keep}, "log": log}
                print(n, name, f"{dt:.1f} s", flush=True)
        data[str(n)] = rec
        out_path.write_text(json.dumps(data, indent=1) + "\n")


if __name__ == "__main__":
    main()
