#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace database of tools/prover_time.py --native: the kernels of the LAST proof in launch order up to the first opening round,
with start offsets, durations and the idle gap in front of each (gaps >= `mingap` us are flagged).  Usage: proof_timeline.py results.db [mingap]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
mingap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
rows = db.execute(f"select {namecol}, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
if len(sys.argv) > 3:
    print("columns:", cols)
steps = [i for i, r in enumerate(rows) if "k_ipa_step" in r[0]]
# the last proof: its 16 step kernels are the last 16; walk back from the first of them to the previous proof's last kernel (a gap > 300 us of host time is not reliable: use the 17th-last step)
first_step = steps[-16]
prev_end = steps[-17] if len(steps) >= 32 else 0
# previous proof ends ~after its last round's kernels: find the first kernel after prev proof's sg/finish by the largest gap between prev_end and first_step
seg = rows[prev_end:first_step]
gaps = [(seg[i + 1][1] - max(r[2] for r in seg[:i + 1]), i) for i in range(len(seg) - 1)]
start_i = prev_end + 1 + max(gaps)[1]
t0 = rows[start_i][1]
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kh::", "").replace("<FqParams>", "").replace("<FpParams>", "").replace("FpParams", "Fp").replace("FqParams", "Fq")[:34]
end_so_far = t0
busy = 0
for row in rows[start_i:first_step]:
    n, s, e = row[:3]
    gap = (s - end_so_far) / 1e3
    flag = "  <-- idle" if gap >= mingap else ""
    print(f"+{(s - t0) / 1e3:8.1f} us  {short(n):34s} {(e - s) / 1e3:7.1f} us   gap {gap:6.1f}{('  q' + str(row[3])) if qcol else ''}{flag}")
    if e > end_so_far:
        busy += e - max(s, end_so_far); end_so_far = e
print(f"until the opening: {(rows[first_step][1] - t0) / 1e3:.0f} us wall, {busy / 1e3:.0f} us with a kernel running")
