#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (kernel-trace run): python tools/dbstats.py file.db [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
tot = sum(r[2] for r in rows)
print("total kernel time %.3f ms" % (tot / 1e6))
for name, cnt, t, mn, mx in rows[:top]:
    print("%-70s %6d %10.3f ms %5.1f%% avg %9.1f us min %8.1f max %9.1f" % (name[:70], cnt, t / 1e6, 100.0 * t / tot, t / cnt / 1e3, mn / 1e3, mx / 1e3))
