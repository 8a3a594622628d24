import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import proof_systems_amd.khip as khip
khip.init(0)
rng = np.random.default_rng(3)
def rs(m):
    s = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 61) - 1); return s
n = 1 << 20
g = khip.srs_generate(0, 0, n)
srs = khip.Srs(0, g)
sc = rs(n)
srs.msm(sc)
t0 = time.perf_counter(); [srs.msm(sc) for _ in range(5)]; dt = (time.perf_counter() - t0) / 5
print(f"kh_msm 2^20 host scalars (PCIe incl.): {1e3*dt:.3f} ms = {n/dt/1e6:.1f} Mscalar/s")
n16 = 1 << 16
srs16 = khip.Srs(0, g[:n16]); srs16.set_lagrange(16, g[:n16])
cols = rs(15 * n16).reshape(15, n16, 4)
srs16.msm_batch(cols, basis=16)
t0 = time.perf_counter(); [srs16.msm_batch(cols, basis=16) for _ in range(5)]; dt = (time.perf_counter() - t0) / 5
print(f"kh_msm_batch 15 x 2^16 host scalars: {1e3*dt:.3f} ms")
x = rs(19 * n16).reshape(19, n16, 4)
khip.ntt(0, x, 16, True)
t0 = time.perf_counter(); [khip.ntt(0, x, 16, True) for _ in range(5)]; dt = (time.perf_counter() - t0) / 5
print(f"kh_ntt 19 x 2^16 host buffers (H2D + D2H + numpy copy): {1e3*dt:.3f} ms")
