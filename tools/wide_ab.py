#!/usr/bin/env python3
"""A/B of the two window-table sets on ONE handle and ONE box: the 2^20-point Vesta MSM through the narrow tables (c = 16, 2^15 buckets) and
through the wide ones (c = 20, 2^19 buckets, two-plane lazy reduction), interleaved.  kh_msm_set_wide_min_n switches per call.
Prints the synchronous per-phase HIP-event times (median of 7) of both, then `alternations` rounds of the pipelined loop of bench.py
(depth 2, `steps` MSMs) for each, and checks that both paths return the same point.  Usage: wide_ab.py [log_n] [steps] [alternations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
alts = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 1 << log_n
khip.init(0)
khip.set_wide_min_n(n)
srs = khip.Srs.create(khip.VESTA, n)
rng = np.random.default_rng(1234)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 61) - 1)
d = khip.DevBuf(sc.nbytes).upload(sc)
MODES = {"narrow c=16": 0, "wide c=20": n}


def phases(reps=7):
    acc = {}
    for _ in range(2):
        out = srs.msm_batch_dev(d.ptr, n, 1)
    for _ in range(reps):
        out = srs.msm_batch_dev(d.ptr, n, 1)
        for name, ms in khip.last_timings():
            acc.setdefault(name, []).append(ms)
    return {k: float(np.median(v)) for k, v in acc.items()}, out


def pipelined(depth=2):
    khip.set_phase_timers(False)                                   # as bench.py's timed regions
    khip.sync()
    t0 = time.perf_counter()
    pending = []
    for _ in range(steps):
        pending.append(srs.msm_submit(d.ptr, n, 1))
        if len(pending) >= depth:
            srs.msm_wait(pending.pop(0))
    while pending:
        srs.msm_wait(pending.pop(0))
    khip.sync()
    khip.set_phase_timers(True)
    return (time.perf_counter() - t0) / steps


res = {}
for name, thr in MODES.items():
    khip.set_wide_min_n(thr)
    ph, out = phases()
    res[name] = out
    tot = sum(v for k, v in ph.items() if not k.startswith("k_"))
    print(f"{name:12s} sync us: " + "  ".join(f"{k} {v * 1e3:.0f}" for k, v in ph.items()) + f"  | sum {tot * 1e3:.0f}")
a, b = res["narrow c=16"], res["wide c=20"]
assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "the two table sets disagree"
print("same point from both table sets: OK")
rates = {k: [] for k in MODES}
for r in range(alts):
    for name, thr in MODES.items():
        khip.set_wide_min_n(thr)
        pipelined()                                            # re-warm this mode's workspaces / clocks
        t = pipelined()
        rates[name].append(n / t / 1e6)
for name in MODES:
    v = rates[name]
    print(f"{name:12s} pipelined depth 2, {steps} steps, Mscalar/s: " + " ".join(f"{x:.0f}" for x in v) + f"  | median {np.median(v):.0f}")
for depth in (1, 3, 4):
    for name, thr in MODES.items():
        khip.set_wide_min_n(thr)
        pipelined(depth); t = pipelined(depth)
        print(f"{name:12s} depth {depth}: {n / t / 1e6:.0f} Mscalar/s")
