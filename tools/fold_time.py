import time, numpy as np, sys
sys.path.insert(0, '.')
from proof_systems_amd import khip
khip.init(0)
n = 1 << 15
g = khip.srs_generate(0, 0, 2 * n)
u = np.array([12345678901234567890123456789 & (2**64 - 1), 12345678901234567890123456789 >> 64, 0, 0], dtype=np.uint64)
for name, f in (("generic", lambda: khip.ipa_fold_points(0, g[:n], g[n:], u)), ("endo", lambda: khip.ipa_fold_points_endo(0, g[:n], g[n:], (1 << 127) | 0x1234567890abcdef1234567))):
    f(); t = time.perf_counter()
    for _ in range(10): f()
    print(name, "n=2^15 fold ms", (time.perf_counter() - t) * 100)
