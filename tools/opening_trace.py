#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace database of tools/ipa_time.py: every kernel of the LAST complete opening (from its first k_ipa_step to its last kernel) in
start order -- offset, duration, end offset, queue, name -- with the rebase's side-stream kernels (k_rb_*, k_precompute) marked.  Shows what the background
materialisation overlaps with and which of the round's own kernels stretch underneath it.  Usage: opening_trace.py results.db [rounds]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f"{namecol}, start, end" + (f", {qcol}" if qcol else ", 0")
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
steps = [i for i, r in enumerate(rows) if "k_ipa_step" in r[0]]
# ipa_time.py ends with ONE extra round after the repetitions: drop it
steps = steps[:-1]
first = steps[-rounds]
t0 = rows[first][1]
short = lambda n: n.split("(")[0].replace("void ", "").replace("kh::", "").replace("<FqParams>", "").replace("<FpParams>", "").replace("FpParams", "Fp").replace("FqParams", "Fq")[:30]
last_end = max(r[2] for r in rows[first:steps[-1] + 12])
rn = 0
for i in range(first, len(rows)):
    n, s, e, q = rows[i]
    if s > last_end:
        break
    if "k_ipa_step" in n:
        rn += 1
        print(f"--- round {rn}")
    side = "  <== side stream" if ("k_rb_" in n or "k_precompute" in n) else ""
    print(f"+{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  -> {(e - t0) / 1e3:9.1f}   q{q}  {short(n)}{side}")
