#!/usr/bin/env python3
"""Kernel timeline of ONE opening round from a rocprofv3 --kernel-trace database: finds the last k_ipa_step launch but 3
and prints every kernel until the next one (start offset, duration, gap to the previous kernel's end).
Usage: tools/trace_round.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "k_ipa_step" in r[0]]
if len(marks) < 5:
    print("no opening rounds in the trace"); sys.exit(0)
a, b = marks[-4], marks[-3]
t0 = rows[a][1]; prev_end = t0
tot = 0
for name, s, e in rows[a:b]:
    short = name.split("(")[0].replace("void ", "").replace("kh::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {short[:60]}")
    prev_end = max(prev_end, e); tot += e - s
print(f"round: {(rows[b][1] - t0) / 1e3:.1f} us wall, {tot / 1e3:.1f} us of kernels, {b - a} launches")
