import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import proof_systems_amd.khip as khip
from oracle import cref
khip.init(0)
n = 1 << 20
g = khip.srs_generate(0, 0, n)
srs = khip.Srs(0, g)
rng = np.random.default_rng(5)
R = np.array([0x34786d38fffffffd, 0x992c350be41914ad, 0xffffffffffffffff, 0x3fffffffffffffff], dtype=np.uint64)
one = np.tile(R, (n, 1))
s = rng.integers(0, 1 << 64, size=(1, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 61) - 1)
same = np.tile(s, (n, 1))
small = np.zeros((n, 4), np.uint64); small[:, 0] = rng.integers(0, 1 << 20, n).astype(np.uint64)   # Montgomery limbs tiny -> canonical values random; use mont=False below
for name, sc, mont in (("all ones", one, True), ("all equal random", same, True), ("20-bit canonical scalars", small, False)):
    buf = khip.DevBuf(sc.nbytes).upload(sc)
    srs.msm_batch_dev(buf.ptr, n, 1, mont=mont)
    t0 = time.perf_counter(); out, inf = srs.msm_batch_dev(buf.ptr, n, 1, mont=mont); dt = time.perf_counter() - t0
    want, winf = cref.msm(0, g, sc, scalars_mont=mont, threads=64)
    print(f"{name:28s} {1e3*dt:8.3f} ms  match={bool(inf[0]) == winf and (winf or np.array_equal(out[0], want))}  phases:", " ".join(f"{k}={v:.3f}" for k, v in khip.last_timings()))
    buf.free()
