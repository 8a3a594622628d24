#!/usr/bin/env python3
"""The 15 witness-column commitments of the benchmark circuit (n - 10 ones, 7 zeros, 3 random per column: kimchi/src/bench.rs:106) as
one batched MSM over the Lagrange basis, device resident: wall time and the library's per-phase HIP-event timings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
from proof_systems_amd import prover
khip.init(0)
logn = 16; n = 1 << logn
srs = khip.Srs.create(khip.VESTA, n); srs.compute_lagrange(logn)
F = prover.Fld(khip.FP)
rng = np.random.default_rng(1)
w = np.tile(F.limbs(1), (15, n, 1)); w[:, n - 10:n - 3] = 0
w[:, n - 3:] = F.limbs_many(F.rand_many(rng, 45)).reshape(15, 3, 4)
d = khip.DevBuf(w.nbytes).upload(w)
uni = rng.integers(0, 1 << 64, size=(15 * n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64((1 << 61) - 1)
du = khip.DevBuf(uni.nbytes).upload(uni)
for name, buf in (("bench witness", d), ("uniform", du)):
    for _ in range(3): srs.msm_batch_dev(buf.ptr, n, 15, basis=logn)
    ts = []
    for _ in range(10):
        khip.sync(); t0 = time.perf_counter(); srs.msm_batch_dev(buf.ptr, n, 15, basis=logn); ts.append(time.perf_counter() - t0)
    print(f"{name}: 15 x 2^16 commit {1e3 * min(ts):.3f} ms   phases:", "  ".join(f"{k} {v:.3f}" for k, v in khip.last_timings()))
