#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace) into a per-kernel stats CSV (name, calls, total/avg/min/max ns, %)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for r in rows:
    print(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f}')
