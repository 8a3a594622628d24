"""Phase times of k_sort_fused (block 0's wall-clock stamps between its grid barriers) for a k = 2 batch of 2^16-point MSMs.
Usage: KH_FUSED_DEBUG=1 python tools/fused_phases.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
khip.init(0)
srs = khip.Srs.create(khip.VESTA, 1 << 16)
rng = np.random.default_rng(3)
sc = rng.integers(0, 1 << 62, size=(2, 1 << 16, 4), dtype=np.uint64)
for _ in range(6):
    t = time.perf_counter(); r = srs.msm_batch(sc); print(f"{1e6 * (time.perf_counter() - t):.0f} us", file=sys.stderr)
