#!/usr/bin/env python3
"""Device-resident NTT / LDE timings (HIP events on the library stream) with algorithmic GB/s
(64 B per element in place, 288*n per LDE column) and Montgomery products per second."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip

khip.init(0)
rng = np.random.default_rng(1)


def rs(m):
    s = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 61) - 1); return s


def timed(fn, reps=10):
    fn(); khip.sync()
    ts = []
    for _ in range(reps):
        fn(); khip.sync()
        ts.append(sum(ms for _, ms in khip.last_timings()))
    return float(np.median(ts))


print(f"{'op':28s} {'ms':>9s} {'alg GB/s':>10s} {'%HBM':>7s} {'Gmul/s':>8s}")
BENCH_SHAPES = "--bench-shapes" in sys.argv       # only the two shapes bench.py's ntt_kernels block quotes (the PMC profile of that block)
for logn, batch in ([(16, 19)] if BENCH_SHAPES else [(12, 64), (16, 1), (16, 19), (18, 1), (19, 1), (20, 1), (22, 1)]):
    n = 1 << logn
    buf = khip.DevBuf(batch * n * 32).upload(rs(batch * n))
    for inv in (True,):
        ms = timed(lambda: khip.ntt_dev(khip.FP, buf, logn, inv, batch))
        gbs = 64.0 * n * batch / (ms * 1e-3) / 1e9
        muls = (0.5 * logn + 1.0) * n * batch / (ms * 1e-3) / 1e9
        print(f"{'intt 2^%d x%d' % (logn, batch):28s} {ms:9.4f} {gbs:10.1f} {100 * gbs / 8000:7.2f} {muls:8.1f}")
    buf.free()
for logn, logb, batch in ([(16, 3, 16)] if BENCH_SHAPES else [(16, 3, 16), (16, 3, 1), (12, 3, 16)]):
    n = 1 << logn
    src = khip.DevBuf(batch * n * 32).upload(rs(batch * n))
    dst = khip.DevBuf(batch * (n << logb) * 32)
    ms = timed(lambda: khip.lde_dev(khip.FP, src, logn, logb, dst, batch))
    gbs = (32.0 + 32.0 * (1 << logb)) * n * batch / (ms * 1e-3) / 1e9
    muls = (0.5 * logn + 2.0) * (n << logb) * batch / (ms * 1e-3) / 1e9
    print(f"{'lde 2^%d->2^%d x%d' % (logn, logn + logb, batch):28s} {ms:9.4f} {gbs:10.1f} {100 * gbs / 8000:7.2f} {muls:8.1f}")
    src.free(); dst.free()
