#!/usr/bin/env python3
"""Summarises a rocprofv3 kernel trace CSV: python tools/trace_summary.py trace.csv [kernel-substring]
Prints per-kernel totals over the LAST third of the dispatches (the last repetition of a 3-call run) and, for
the named kernel, its durations and the idle gaps between its launches."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "vslam" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 3
last = rows[-n:]
t0, t1 = int(last[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in last)
print("span %.3f ms over %d dispatches" % ((t1 - t0) / 1e6, n))
tot = defaultdict(lambda: [0, 0.0])
for r in last:
    k = r["Kernel_Name"].split("(")[0][-60:]
    tot[k][0] += 1
    tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %5d calls %9.1f us  avg %7.1f" % (k, c, us, us / c))
if len(sys.argv) > 2:
    sel = [r for r in last if sys.argv[2] in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(sel, sel[1:])]
    print("durations us:", [round(x) for x in d])
    print("gaps us:", [round(x) for x in gaps])
    print("sum durations %.1f us, sum gaps %.1f us" % (sum(d), sum(gaps)))
