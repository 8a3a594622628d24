#!/usr/bin/env python3
"""Kernel timeline of the LAST proof in a rocprofv3 kernel trace of tools/prover_time.py: name, start (us from the proof's first
kernel), duration, idle gap before it -- where the GPU waits for the host (transcript, synchronous results) between the phases.
Usage: rocprofv3 --kernel-trace -d DIR -o pp -- python tools/prover_time.py 16;  tools/prover_timeline.py DIR/pp_results.db [--opening]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {namecol},start,end from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"kh::", "", n); return n.split("(")[0][:36]
ipa = [i for i, r in enumerate(rows) if "k_ipa_step" in r[0]]
a, b = ipa[-17], ipa[-16]                       # last step of the previous proof, first step of the last one
seg = rows[a:b + 1]
g, i = max((seg[k + 1][1] - seg[k][2], k) for k in range(len(seg) - 1))
start = a + i + 1
end = len(rows) if "--opening" in sys.argv else b + 1
tb = rows[start][1]; prev = None; ksum = 0.0; idle = 0.0
for n, s, e in rows[start:end]:
    gap = (s - prev) / 1e3 if prev else 0.0
    ksum += (e - s) / 1e3; idle += max(gap, 0.0)
    print(f"{short(n):38s} t={(s - tb) / 1e3:9.1f} dur={(e - s) / 1e3:8.1f} gap={gap:8.1f}")
    prev = e
print(f"kernel sum {ksum:.1f} us, idle {idle:.1f} us, span {(rows[end - 1][2] - tb) / 1e3:.1f} us")
