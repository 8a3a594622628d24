#!/usr/bin/env python3
"""Per-phase device timings of the prover op-list pieces (debug aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
khip.init(0)
n = 1 << 16
rng = np.random.default_rng(3)
def rs(m):
    s = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 61) - 1); return s
g = khip.srs_generate(0, 0, n)
srs = khip.Srs(0, g)
srs.set_lagrange(16, g)
F_R = np.array([0x34786d38fffffffd, 0x992c350be41914ad, 0xffffffffffffffff, 0x3fffffffffffffff], dtype=np.uint64)
wit = np.tile(F_R, (15, n, 1)); wit[:, n - 10: n - 3] = 0; wit[:, n - 3:] = rs(45).reshape(15, 3, 4)
d_wit = khip.DevBuf(wit.nbytes).upload(wit)
d_zt = khip.DevBuf(8 * n * 32).upload(rs(8 * n))
for name, fn in [("witness x15 (lagrange)", lambda: srs.msm_batch_dev(d_wit.ptr, n, 15, basis=16)),
                 ("z,t x8 (monomial)", lambda: srs.msm_batch_dev(d_zt.ptr, n, 8)),
                 ("single 2^16", lambda: srs.msm_batch_dev(d_zt.ptr, n, 1))]:
    fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name:26s} wall {1e3*dt:7.3f} ms  phases:", " ".join(f"{k}={v:.3f}" for k, v in khip.last_timings()))
for m in (32768, 4098, 514, 66, 10, 3):
    pts = np.stack([g[:m], g[n // 2: n // 2 + m]]); sc = rs(2 * m).reshape(2, m, 4)
    khip.msm_points_batch(0, pts, sc)
    t0 = time.perf_counter(); khip.msm_points_batch(0, pts, sc); dt = time.perf_counter() - t0
    print(f"ipa round m={m:6d} x2       wall {1e3*dt:7.3f} ms  phases:", " ".join(f"{k}={v:.3f}" for k, v in khip.last_timings()))
