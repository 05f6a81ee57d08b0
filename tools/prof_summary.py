#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / counter collection) into small summaries that can be committed under
profiles/.  Usage: prof_summary.py <rocprof_output_dir> <out_prefix>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    src, out = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
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
        for r in csv.DictReader(open(cc[0])):
            k = short(r.get("Kernel_Name", r.get("Kernel_name", "?")))
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        res = {k: {c: {"dispatches": v[0], "sum": v[1], "mean_per_dispatch": v[1] / max(v[0], 1)} for c, v in d.items()} for k, d in agg.items()}
        json.dump(res, open(out + "_pmc.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
