#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / counter collection) into small summaries that can be committed under
profiles/.  Usage: prof_summary.py <rocprof_output_dir> <out_prefix> [--skip-frac F]
--skip-frac F: per kernel name, drop the first F of its dispatches in time order before averaging (tools/profile_pass.py: F = 0.25 of a 2048-token
pass removes the decode-step kernels' ramp-up to the steady-state context sawtooth); then the kernel stats come from the per-dispatch kernel
trace (*kernel_trace.csv), not from rocprofv3's own *kernel_stats.csv."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def trace_stats(src, out, skip):
    tr = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not tr:
        return False
    by = defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        by[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows = []
    for k, v in by.items():
        v.sort()
        v = v[int(len(v) * skip):]
        d = [(e - s) / 1e3 for s, e in v]
        if d:
            rows.append((k, len(d), sum(d) / 1e3, sum(d) / len(d), min(d), max(d)))
    tot = sum(r[2] for r in rows) or 1.0
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in sorted(rows, key=lambda r: -r[2]):
            f.write(f"{r[0]},{r[1]},{r[2]:.3f},{r[3]:.3f},{r[4]:.3f},{r[5]:.3f},{100 * r[2] / tot:.2f}\n")
    return True


def main():
    src, out = sys.argv[1], sys.argv[2]
    skip = float(sys.argv[sys.argv.index("--skip-frac") + 1]) if "--skip-frac" in sys.argv else 0.0
    stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if skip > 0 and trace_stats(src, out, skip):
        stats = []
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(out + "_kernel_stats.csv", "w") as f:
            f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
            for r in rows:
                f.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.3f},"
                        f"{float(r['MinNs']) / 1e3:.3f},{float(r['MaxNs']) / 1e3:.3f},{r['Percentage']}\n")
    cc = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    if cc:
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        per = defaultdict(list)                                  # (kernel, counter) -> values in dispatch order
        for r in csv.DictReader(open(cc[0])):
            k = short(r.get("Kernel_Name", r.get("Kernel_name", "?")))
            per[(k, r["Counter_Name"])].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
        for (k, cname), vals in per.items():
            vals.sort()
            vals = vals[int(len(vals) * skip):]
            a = agg[k][cname]
            a[0] += len(vals)
            a[1] += sum(v for _, v in vals)
        res = {k: {c: {"dispatches": v[0], "sum": v[1], "mean_per_dispatch": v[1] / max(v[0], 1)} for c, v in d.items()} for k, d in agg.items()}
        json.dump(res, open(out + "_pmc.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
