"""Times the device-resident opening loop (kh_ipa_*) at 2^16 on Vesta: per-round wall time and the MSM phases."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proof_systems_amd import khip
khip.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << logn
rng = np.random.default_rng(1)
def rs(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
srs = khip.Srs.create(0, n)
U = khip.srs_generate(0, 1 << 21, 1)[0]
a = rs(n); b = rs(n); r = rs(2)
chals = [int.from_bytes(rng.bytes(16), "little") for _ in range(logn)]
for rep in range(3):
    khip.sync(); t0 = time.perf_counter()
    op = khip.IpaOpening(srs, a, b, U)
    t1 = time.perf_counter()
    per = []
    for ch in chals:
        ta = time.perf_counter()
        op.round_lr(r[0], r[1]); tb = time.perf_counter()
        op.round_fold(ch); per.append((tb - ta, time.perf_counter() - tb))
    t2 = time.perf_counter()
    op.finish(); t3 = time.perf_counter()
    op.free()
    print(f"rep {rep}: begin {1e3*(t1-t0):.3f} ms, rounds {1e3*(t2-t1):.3f} ms, finish {1e3*(t3-t2):.3f} ms, total {1e3*(t3-t0):.3f} ms")
print("per round lr/fold ms:", " ".join(f"{1e3*x:.3f}/{1e3*y:.3f}" for x, y in per))
op = khip.IpaOpening(srs, a, b, U)
op.round_lr(r[0], r[1]); khip.sync()
print("round 0 MSM phases:", [(k, round(v, 4)) for k, v in khip.last_timings()])
