#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share, min, max) of a rocprofv3 --kernel-trace run from its results
database, in the column layout of rocprofv3's own kernel_stats.csv.
usage: kernel_stats_from_db.py <results.db> > profiles/..._kernel_stats.csv"""
import csv
import sqlite3
import subprocess
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
                  f"max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1


def demangle(n):
    n = n[:-3] if n.endswith(".kd") else n
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True, timeout=10).stdout.strip() or n
    except (OSError, subprocess.SubprocessError):
        return n


w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, calls, tot, avg, mn, mx in rows:
    w.writerow([demangle(name), calls, int(tot), round(avg, 2), round(100.0 * tot / total, 2), int(mn), int(mx)])
