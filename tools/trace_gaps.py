#!/usr/bin/env python3
"""Where a chain of small dependent kernels spends its time: from a rocprofv3 --kernel-trace CSV, per kernel (name, grid) the mean DURATION
(End - Start) and the mean GAP to the previous kernel on the same queue (Start - previous End), over the last `--last` dispatches (the timed
loop, not the setup).  usage: trace_gaps.py DIR [--last N]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if last:
    rows = rows[-last:]
agg = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = {}
t_first, t_last = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = 0.0
for r in rows:
    q = r.get("Queue_Id", "0")
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = (name[:56], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg[key]; a[0] += 1; a[1] += (e - s) / 1e3
    busy += (e - s) / 1e3
    if q in prev_end:
        a[2] += max(0, s - prev_end[q]) / 1e3
    prev_end[q] = e
span = (t_last - t_first) / 1e3
print(f"{len(rows)} dispatches, span {span:.1f} us, kernels busy {busy:.1f} us ({100 * busy / span:.1f} %), gaps {span - busy:.1f} us")
print(f"{'kernel':56s} {'grid':>8s} {'wg':>5s} {'n':>6s} {'dur_us':>9s} {'gap_us':>8s} {'share':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:30]:
    print(f"{k[0]:56s} {k[1]:>8s} {k[2]:>5s} {v[0]:6d} {v[1] / v[0]:9.2f} {v[2] / v[0]:8.2f} {100 * (v[1] + v[2]) / span:5.1f}%")
