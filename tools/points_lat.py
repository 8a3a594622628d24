"""Latency of ad-hoc-basis MSMs (kh_msm_points: the plain per-window path; verifier / small-SRS use)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from proof_systems_amd import khip
khip.init(0)
rng = np.random.default_rng(1)
def rs(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
g = khip.srs_generate(0, 0, 1 << 14)
for n in (16, 64, 512, 4096, 16384):
    sc = rs(n)
    ts = []
    for _ in range(9):
        t = time.perf_counter(); khip.msm_points(0, g[:n], sc); ts.append(time.perf_counter() - t)
    print(f"n={n}: {1e3*np.median(ts):.3f} ms", [(a, round(b, 3)) for a, b in khip.last_timings()])
