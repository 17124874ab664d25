#!/usr/bin/env python3
"""tools/calibrate_pmc.py <dir with ubench.json + ubench_pmc_{FETCH,WRITE}_SIZE.csv> <out.json>

What rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB) report per access pattern, against the exact byte counts of
tools/ubench_lines' kernels: bytes the counter shows per access, and the factor that turns the counter into the
bytes the kernel asked for.  (MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide streaming reads: x 2.)"""
import collections
import csv
import json
import sys


def main() -> None:
    src, out = sys.argv[1], sys.argv[2]
    tests = json.load(open(f"{src}/ubench.json"))["results"]
    accesses = 64 << 20
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f"{src}/ubench_pmc_{counter}.csv")):
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
        per[counter] = acc
    rows = []
    sizes = [t["working_set_mb"] for t in tests if t["test"] == "gather64"]
    for name in ("k_stream_copy", "k_gather64", "k_gather16", "k_gather8", "k_scatter8", "k_scatter16", "k_atom_add64", "k_atom_add32"):
        f, w = per["FETCH_SIZE"].get(name, []), per["WRITE_SIZE"].get(name, [])
        reps = 5
        for s in range(len(f) // reps):
            fv = sorted(f[s * reps:(s + 1) * reps])[reps // 2]
            wv = sorted(w[s * reps:(s + 1) * reps])[reps // 2]
            rec = {"kernel": name, "fetch_counter_bytes": fv, "write_counter_bytes": wv}
            if name == "k_stream_copy":
                rec.update(bytes_read=2 << 30, bytes_written=2 << 30, fetch_factor=(2 << 30) / fv, write_factor=(2 << 30) / wv)
            else:
                rec.update(working_set_mb=sizes[s], accesses=accesses, fetch_bytes_per_access=fv / accesses, write_bytes_per_access=wv / accesses)
            rows.append(rec)
    summary = {
        "streaming 16 B/lane reads": "FETCH_SIZE shows half the bytes (x 2, as the guide says); WRITE_SIZE exact",
        "random reads of 8 / 16 / 64 B": "FETCH_SIZE shows 64 B per access once the set exceeds L2 (x 1: a random access is one 64-byte request)",
        "random 8 / 16 B stores": "WRITE_SIZE shows ~32 B per store (x 1)",
        "device-scope atomics": "WRITE_SIZE shows 32 B per atomic, FETCH_SIZE nothing (executed memory side)",
    }
    json.dump({"summary": summary, "rows": rows, "ceilings": tests}, open(out, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
