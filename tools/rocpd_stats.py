#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into the table
`rocprofv3 --stats` reports: per-kernel calls, total/avg/min/max duration (ns), percentage.
Usage: tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = ", avg(vgpr_count), avg(lds_size)" if "vgpr_count" in cols and "lds_size" in cols else ", null, null"
rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start){extra} "
                  f"from kernels group by {namecol} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,LDS"]
for name, calls, tot, avg, mn, mx, vg, lds in rows:
    lines.append(f'"{name}",{calls},{tot},{avg:.1f},{mn},{mx},{100.0 * tot / total:.2f},{"" if vg is None else vg},{"" if lds is None else lds}')
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
