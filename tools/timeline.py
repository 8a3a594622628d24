#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace rocpd database: how much of the steady-state wall time has a k_accumulate running,
how much has two overlapping, and what else runs in the gaps.  Usage: tools/timeline.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
acc = [(s, e) for n, s, e in rows if "k_accumulate29" in n or "k_acc_wide29" in n] or [(s, e) for n, s, e in rows if "k_accumulate" in n]
acc = acc[4:-2]                                            # steady state
t0, t1 = acc[0][0], acc[-1][1]
ev = sorted([(s, 1) for s, e in acc] + [(e, -1) for s, e in acc])
cur, last, cov = 0, t0, {0: 0, 1: 0, 2: 0, 3: 0}
for t, d in ev:
    cov[min(cur, 3)] += t - last; last = t; cur += d
tot = t1 - t0
print(f"steady window {tot/1e6:.3f} ms over {len(acc)} accumulate launches = {tot/1e6/len(acc):.3f} ms each")
for k in cov: print(f"  {k} accumulate kernels running: {100*cov[k]/tot:.1f} %")
print(f"  mean accumulate duration {sum(e-s for s,e in acc)/len(acc)/1e6:.3f} ms")
gaps = []
cur = 0; last = t0
for t, d in ev:
    if cur == 0 and t > last: gaps.append((last, t))
    last = t; cur += d
others = {}
for n, s, e in rows:
    if "k_accumulate" in n or "k_acc_wide29" in n: continue
    for gs, ge in gaps:
        o = min(e, ge) - max(s, gs)
        if o > 0: others[n.split("(")[0][-40:]] = others.get(n.split("(")[0][-40:], 0) + o
for n, o in sorted(others.items(), key=lambda x: -x[1])[:8]: print(f"  in gaps: {n:40s} {o/1e6:.3f} ms")
