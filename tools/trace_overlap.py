#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace (tuning aid): device busy fraction, mean concurrency, and per-queue gaps between
consecutive kernels.  Usage: trace_overlap.py <rocprof_output_dir> [t_from_frac t_to_frac]"""
import csv, glob, os, sys
from collections import defaultdict

src = sys.argv[1]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
path = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
T0, T1 = rows[0][0], max(r[1] for r in rows)
lo, hi = T0 + (T1 - T0) * f0, T0 + (T1 - T0) * f1
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
print(f"{len(rows)} kernels in window of {(hi - lo) / 1e6:.1f} ms")
# busy union + concurrency
ev = []
for s, e, *_ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; conc_time = defaultdict(int); cur = 0; last = ev[0][0]
for t, d in ev:
    if cur > 0: busy += t - last
    conc_time[cur] += t - last
    cur += d; last = t
span = rows[-1][1] - rows[0][0]
tot = sum(e - s for s, e, *_ in rows)
print(f"span {span / 1e6:.1f} ms, busy (union) {busy / 1e6:.1f} ms = {busy / span:.1%}, sum of durations {tot / 1e6:.1f} ms, mean concurrency while busy {tot / busy:.2f}")
print("time at concurrency level:", {k: f"{v / span:.1%}" for k, v in sorted(conc_time.items())})
# per-queue gaps
byq = defaultdict(list)
for s, e, q, st, n in rows:
    byq[(q, st)].append((s, e, n))
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:8]:
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    gaps_s = sorted(gaps)
    dur = sum(e - s for s, e, _ in lst)
    print(f"queue/stream {q}: {len(lst)} kernels, busy {dur / 1e6:.1f} ms, gaps: median {gaps_s[len(gaps_s) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, "
          f"p90 {gaps_s[int(len(gaps_s) * 0.9)] / 1e3:.2f} us, total gap {sum(g for g in gaps if g > 0) / 1e6:.1f} ms")
# per-kernel mean duration
agg = defaultdict(lambda: [0, 0])
for s, e, q, st, n in rows:
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {n:45s} {c:8d} x {t / c / 1e3:8.2f} us = {t / 1e6:9.1f} ms")
