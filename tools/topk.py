#!/usr/bin/env python3
"""Prints the per-kernel summary of a rocprofv3 rocpd database (kernel-trace run)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("%-90s %7s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (limit,)):
    print("%-90s %7d %12.1f %10.2f %6.2f" % (name[:90], calls, total / 1e3, avg / 1e3, pct))
