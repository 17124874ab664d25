#!/usr/bin/env python3
"""tools/summarize_pmc_kernels.py <dir with kernel_stats.csv + pmc*.csv from tools/profile_cmd.sh> <out.json> [name regex]

Per kernel of the library: dispatches, average duration (kernel trace), and per-dispatch averages of every counter; derived:
HBM bytes per dispatch = c x FETCH_SIZE + WRITE_SIZE (KiB units x 1024) with BOTH corrections: c = 2 as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (right for wide streaming reads) and c = 1 (right for kernels whose
fetches are random 64-byte lines: profiles/r03/ubench_ceilings_and_pmc_calibration.json), GB/s against the 8 TB/s peak, VALU wave-instructions/s against
256 CUs x 4 SIMDs x 0.5 wave-instructions per cycle x 2.4 GHz (bench.py's VALU_PEAK_WAVE_INSTR_S).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

HBM_PEAK = 8.0e12
VALU_PEAK = 256 * 4 * 0.5 * 2.4e9


def short(name: str) -> str:
    m = re.search(r"(k_\w+(<[^>(]*>)?)", name)
    return m.group(1) if m else name


def main() -> None:
    src, out = sys.argv[1], sys.argv[2]
    want = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
    dur = {}
    with open(os.path.join(src, "kernel_stats.csv")) as fh:
        for row in csv.DictReader(fh):
            if "anonymous namespace" in row["Name"]:
                k = short(row["Name"])
                calls, total = int(row["Calls"]), float(row["TotalDurationNs"])
                c0, t0 = dur.get(k, (0, 0.0))
                dur[k] = (c0 + calls, t0 + total)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "pmc*.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "anonymous namespace" in row["Kernel_Name"]:
                    per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    table = {}
    for k, (calls, total) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        if want is not None and not want.search(k):
            continue
        avg_s = total / calls * 1e-9
        rec = {"dispatches": calls, "avg_us": round(avg_s * 1e6, 2), "total_ms": round(total * 1e-6, 3)}
        cs = {c: sum(v) / len(v) for c, v in per.get(k, {}).items()}
        rec["counters_per_dispatch"] = {c: round(v, 1) for c, v in sorted(cs.items())}
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            hbm = (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0
            rec["hbm_bytes_per_dispatch"] = round(hbm)
            rec["hbm_bytes_per_dispatch_fetch_x1"] = round((cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0)
            rec["hbm_GBps"] = round(hbm / avg_s * 1e-9, 1)
            rec["hbm_GBps_fetch_x1"] = round((cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0 / avg_s * 1e-9, 1)
            rec["frac_of_hbm_peak"] = round(hbm / avg_s / HBM_PEAK, 4)
        if "SQ_INSTS_VALU" in cs:
            rec["valu_wave_instr_per_s"] = round(cs["SQ_INSTS_VALU"] / avg_s)
            rec["frac_of_valu_peak"] = round(cs["SQ_INSTS_VALU"] / avg_s / VALU_PEAK, 4)
        if cs.get("SQ_ACTIVE_INST_LDS") and "SQ_LDS_BANK_CONFLICT" in cs:
            rec["lds_conflict_share_of_lds_cycles"] = round(cs["SQ_LDS_BANK_CONFLICT"] / cs["SQ_ACTIVE_INST_LDS"], 4)
        if cs.get("SQ_BUSY_CYCLES") and "SQ_WAVE_CYCLES" in cs:
            rec["wave_cycles_over_busy_cycles"] = round(cs["SQ_WAVE_CYCLES"] / cs["SQ_BUSY_CYCLES"], 2)
        table[k] = rec
    with open(out, "w") as fh:
        json.dump({"peaks": {"hbm_bytes_per_s": HBM_PEAK, "valu_wave_instr_per_s": VALU_PEAK},
                   "note": "durations from the kernel-trace pass, counters from --pmc passes of the same command (kernels run slower "
                           "under counters: rates use the trace's duration); FETCH_SIZE doubled (gfx950) except in the *_fetch_x1 figures (random-line kernels)", "kernels": table}, fh, indent=1)
    for k, r in list(table.items())[:12]:
        print(k, r["avg_us"], r.get("hbm_GBps"), r.get("frac_of_valu_peak"))


if __name__ == "__main__":
    main()
