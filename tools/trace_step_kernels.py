#!/usr/bin/env python3
"""Per-kernel durations of the eager decode step from a rocprofv3 --kernel-trace CSV, split by (kernel, grid, workgroup) so that the
instantiations of one GEMM kernel at different shapes (o-projection vs down-projection) are told apart.  usage: trace_step_kernels.py DIR"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
agg = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k[0][:52]:52s} grid={k[1]:>8s} wg={k[2]:>5s} n={v[0]:6d} avg_us={v[1] / v[0]:9.2f} share={100 * v[1] / tot:5.1f}%")
