#!/usr/bin/env python3
"""tools/summarize_pmc.py <dir with pmc*.csv from tools/profile_d1.sh> <steps+warmup> <workload tag> <out.json>

Per-dispatch averages of every collected counter for the kernels of the d=1 step, and the
HBM traffic of ONE bench step (= one launch of the kernel group the roofline is quoted on: index
build + pair kernels + CSR):
    hbm_bytes_per_launch = sum over the group's dispatches of (2 x FETCH_SIZE + WRITE_SIZE) KiB-units x 1024 / steps
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes for gfx950 (128-B requests
tallied at 64 B); both the corrected and the raw figure are kept.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main() -> None:
    src, steps, tag, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "pmc*.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "anonymous namespace" in name:           # every kernel of the library (not the harness's fills / copies)
                    m = re.search(r"(k_\w+(<[^>(]*>)?)", name)
                    short = m.group(1) if m else name
                    per[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kernels = {}
    fetch_kib = write_kib = 0.0
    for k, cs in sorted(per.items()):
        kernels[k] = {c: {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in sorted(cs.items())}
        fetch_kib += sum(cs.get("FETCH_SIZE", []))
        write_kib += sum(cs.get("WRITE_SIZE", []))
    raw = (fetch_kib + write_kib) * 1024.0 / steps
    corrected = (2.0 * fetch_kib + write_kib) * 1024.0 / steps
    rec = {"workload": tag, "bench_steps_profiled": steps,
           "hbm_bytes_per_launch": corrected, "hbm_bytes_per_launch_uncorrected": raw,
           "note": "launch = one bench step = every dispatch of the library's kernels in the step; "
                   "FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as reported",
           "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({k: rec[k] for k in ("workload", "hbm_bytes_per_launch", "hbm_bytes_per_launch_uncorrected")}))


if __name__ == "__main__":
    main()
