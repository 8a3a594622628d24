#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace database of tools/concurrent_provers.py: over the steady part of the run (the last 60 % of the trace), the fraction of time with at least one
kernel running, the average and the distribution of the number of kernels running at once, and per kernel name the mean duration -- to be read beside the same table of a
one-prover trace (second argument): which kernels stretch when provers share the chip.  Usage: concurrency_trace.py many.db [single.db]"""
import sqlite3
import sys
from collections import defaultdict


def load(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t0 + 0.4 * (t1 - t0)
    return [r for r in rows if r[1] >= lo]


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kh::", "").replace("<FqParams>", "").replace("<FpParams>", "")[:30]


def stats(rows):
    ev = []
    for _, s, e in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    hist = defaultdict(int)
    cur, last = 0, ev[0][0]
    for t, d in ev:
        hist[cur] += t - last
        cur += d; last = t
    total = sum(hist.values())
    per = defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        per[short(n)][0] += 1; per[short(n)][1] += e - s
    return hist, total, per


many = load(sys.argv[1])
hist, total, per = stats(many)
print(f"steady window {total / 1e6:.1f} ms, {len(many)} kernels")
print(f"at least one kernel running: {1 - hist[0] / total:.3f} of the time; mean kernels in flight {sum(k * v for k, v in hist.items()) / total:.2f}")
print("kernels in flight -> share of time: " + "  ".join(f"{k}: {v / total:.3f}" for k, v in sorted(hist.items())))
single = None
if len(sys.argv) > 2:
    _, _, single = stats(load(sys.argv[2]))
print(f"{'kernel':30s} {'calls':>7s} {'mean us':>9s} {'total ms':>9s}" + ("   alone us   stretch" if single else ""))
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    line = f"{n:30s} {c:7d} {t / c / 1e3:9.1f} {t / 1e6:9.2f}"
    if single and n in single:
        a = single[n][1] / single[n][0] / 1e3
        line += f" {a:10.1f} {t / c / 1e3 / a:9.2f}"
    print(line)
