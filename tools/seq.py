#!/usr/bin/env python3
"""Kernel sequence of one blocked Cholesky from a rocprofv3 database: python tools/seq.py file.db [which potf2] [count]"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 12
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
idx = [i for i, r in enumerate(rows) if 'potf2_upper' in r[0]]
i0 = idx[which]
for off in (0, 120):
    for k in range(i0 + off, min(len(rows), i0 + off + cnt)):
        r = rows[k]
        gap = (r[1] - rows[k - 1][2]) / 1e3 if k > 0 else 0.0
        print("%-40s %9.1f us  grid %8d  gap %6.1f" % (r[0][10:48], (r[2] - r[1]) / 1e3, r[3], gap))
    print("...")
