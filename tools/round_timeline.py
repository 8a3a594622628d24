#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace database of a prover run: the kernels of the LAST few opening rounds in launch order with their start offsets, durations
and the idle gap in front of each -- where an opening round's ~340 us go.  Usage: round_timeline.py results.db [rounds]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
steps = [i for i, r in enumerate(rows) if "k_ipa_step" in r[0]]
steps = steps[-(nr + 6):-6] if len(steps) > nr + 6 else steps[-nr:]
short = lambda n: n.split("(")[0].replace("void ", "").replace("kh::", "").replace("<FqParams>", "").replace("<FpParams>", "")[:28]
for si in steps:
    j = si
    t0 = rows[si][1]
    print("round:")
    prev_end = None
    while j < len(rows) and (j == si or "k_ipa_step" not in rows[j][0]):
        n, s, e = rows[j]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"  +{(s - t0) / 1e3:7.1f} us  {short(n):28s} {(e - s) / 1e3:7.1f} us   gap before {gap:6.1f}")
        prev_end = max(prev_end or e, e)
        j += 1
    if j < len(rows):
        print(f"  next round starts at +{(rows[j][1] - t0) / 1e3:.1f} us (host: wait, finish, sponge, submit = {(rows[j][1] - prev_end) / 1e3:.1f} us after the last kernel)")
