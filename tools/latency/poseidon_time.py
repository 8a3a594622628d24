#!/usr/bin/env python3
"""Host Poseidon permutation time through kh_sponge_absorb (scalar mulx path with KH_NO_IFMA=1, AVX-512 IFMA path otherwise). No GPU needed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proof_systems_amd.khip as khip
rng = np.random.default_rng(1)
x = rng.integers(0, 1 << 62, size=(2000, 4), dtype=np.uint64)
s = khip.Sponge(khip.Sponge.FR, khip.VESTA)
s.absorb(x[:100])
best = 1e9
for rep in range(20):
    t = time.perf_counter(); s.absorb(x); best = min(best, time.perf_counter() - t)
print("KH_NO_IFMA=%s: %.2f us per permutation" % (os.environ.get("KH_NO_IFMA", "0"), best / 1000 * 1e6))
