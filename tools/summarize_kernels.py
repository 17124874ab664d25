#!/usr/bin/env python3
"""tools/summarize_kernels.py <dir> <prefix> <out.json> [min_share=0.01]

Per kernel of one profiled command (rocprofv3 --kernel-trace --stats: <prefix>_kernel_stats.csv; separate --pmc runs:
<prefix>_pmc*.csv): calls, total / average duration, and the sum over its dispatches of every collected counter, with
the derived figures the roofline discussion needs — HBM bytes (FETCH_SIZE in 64-byte requests: x 2 for streaming reads,
x 1 for random lines, both given; WRITE_SIZE as reported), VALU wave-instructions per second against the chip's measured
issue ceiling, share of wave cycles spent waiting.  Kernels below min_share of the total time are dropped."""
import collections
import csv
import glob
import json
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(k_\w+(<[^>(]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def main() -> None:
    src, prefix, out = sys.argv[1], sys.argv[2], sys.argv[3]
    min_share = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
    stats = {}
    total_ns = 0.0
    with open(f"{src}/{prefix}_kernel_stats.csv") as fh:
        for row in csv.DictReader(fh):
            k = short(row["Name"])
            s = stats.setdefault(k, {"calls": 0, "total_ms": 0.0})
            s["calls"] += int(row["Calls"])
            s["total_ms"] += float(row["TotalDurationNs"]) * 1e-6
            total_ns += float(row["TotalDurationNs"])
    counters = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in sorted(glob.glob(f"{src}/{prefix}_pmc*.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                counters[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
    res = {}
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"]):
        if s["total_ms"] * 1e6 < min_share * total_ns:
            continue
        c = dict(counters.get(k, {}))
        rec = {"calls": s["calls"], "total_ms": round(s["total_ms"], 4), "avg_ms": round(s["total_ms"] / s["calls"], 4),
               "share_of_gpu_time": round(s["total_ms"] * 1e6 / total_ns, 4), "counters": c}
        sec = s["total_ms"] * 1e-3
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            f, w = c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0
            rec["hbm_bytes_if_streaming"] = 2 * f + w
            rec["hbm_bytes_if_random_lines"] = f + w
            rec["GB/s_if_streaming"] = round((2 * f + w) / sec / 1e9, 1)
            rec["GB/s_if_random_lines"] = round((f + w) / sec / 1e9, 1)
        if "SQ_INSTS_VALU" in c:
            rec["valu_wave_instructions_per_s"] = c["SQ_INSTS_VALU"] / sec
            rec["frac_of_valu_ceiling_4.7e11"] = round(c["SQ_INSTS_VALU"] / sec / 4.7e11, 3)
        if c.get("SQ_WAVE_CYCLES"):
            rec["wave_cycles_waiting_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
            rec["wave_cycles_issuing_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
        res[k] = rec
    json.dump({"gpu_time_ms": round(total_ns * 1e-6, 3), "kernels": res}, open(out, "w"), indent=1)
    for k, r in list(res.items())[:10]:
        print(k, r["avg_ms"], r.get("GB/s_if_streaming"), r.get("GB/s_if_random_lines"), r.get("frac_of_valu_ceiling_4.7e11"), r.get("wave_cycles_waiting_frac"))


if __name__ == "__main__":
    main()
