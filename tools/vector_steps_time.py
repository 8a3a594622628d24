import sys, time
import numpy as np
sys.path.insert(0, '.')
from proof_systems_amd import khip
khip.init(0)
n = 1 << 16
rng = np.random.default_rng(1)
def rs(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
fid = 0
buf = khip.DevBuf(8 * n * 32).upload(rs(8 * n)); q = khip.DevBuf(8 * n * 32); r = khip.DevBuf(n * 32)
def tm(name, f, reps=5):
    f(); khip.sync(); t = time.perf_counter()
    for _ in range(reps): f()
    khip.sync(); print(f"{name}: {1e3 * (time.perf_counter() - t) / reps:.3f} ms")
pts = rs(2)
tm("evaluate_chunks n x2pts", lambda: khip.evaluate_chunks_dev(fid, buf, n, n, 1, pts))
tm("evaluate_chunks 7n x2pts", lambda: khip.evaluate_chunks_dev(fid, buf, 7 * n, n, 7, pts))
tm("batch_inversion n", lambda: khip.batch_inversion_dev(fid, buf, n))
tm("scan mul n", lambda: khip.field_scan_dev(fid, khip.SCAN_MUL, buf, n))
tm("divide_by_linear n", lambda: khip.divide_by_linear_dev(fid, buf, n, pts[0], q))
tm("divide_by_vanishing 8n", lambda: khip.divide_by_vanishing_poly_dev(fid, buf, 8 * n, 16, q, r))
tm("b_init n (2 pts)", lambda: khip.b_init_dev(fid, pts, pts[0], n, q))
polys = [khip.DevBuf(n * 32).upload(rs(n)) for _ in range(45)]
tm("combine_polys 45 x n", lambda: khip.combine_polys_dev(fid, polys, [n] * 45, [1] * 45, pts[0], n, q))
tm("lincomb 20 x n", lambda: khip.poly_lincomb_dev(fid, polys[:20], [n] * 20, rs(20), q, n))
