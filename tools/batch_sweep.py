"""Batched commitments over the resident tables: k MSMs of 2^16 scalars in one call (ms, median of 7) + phases."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from proof_systems_amd import khip
khip.init(0)
rng = np.random.default_rng(1)
def rs(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
n = 1 << 16
srs = khip.Srs.create(0, n)
for k in ([int(x) for x in sys.argv[1:]] or [4, 8, 15, 23, 32]):
    sc = rs(n * k)
    d = khip.DevBuf(sc.nbytes).upload(sc)
    ts = []
    for _ in range(7):
        khip.sync(); t = time.perf_counter(); srs.msm_batch_dev(d.ptr, n, k); ts.append(time.perf_counter() - t)
    print(f"k={k}: {1e3*np.median(ts):.3f} ms ({1e3*np.median(ts)/k:.3f}/MSM)", [(a, round(b, 3)) for a, b in khip.last_timings()])
    d.free()
