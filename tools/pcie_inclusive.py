#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (what a Rust caller of GpuSrs / the ark-poly patch pays; never `value` of bench.py):
the 2^20 MSM from host scalars, the 15-column witness commitment, and the transforms of one proof -- 15 interpolations and 16 eightfold extensions at
2^16 -- called the three ways a caller can: one column per call from ONE thread, one batched call, and one column per call from 15 / 16 threads at
once (the reference's own pattern: par_iter in prover.rs:370-381 and constraints.rs:488-494).  Host memory is pageable (numpy arrays = Rust Vecs)."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proof_systems_amd.khip as khip  # noqa: E402

khip.init(0)
rng = np.random.default_rng(3)


def rs(m):
    s = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 61) - 1)
    return s


def best(f, reps=5):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)


def threaded(fs):
    """run the callables at once, one thread each (threads exist before the clock starts, as a rayon pool does); best of 5"""
    bar = threading.Barrier(len(fs) + 1); done = threading.Barrier(len(fs) + 1)
    reps = 6

    def w(f):
        for _ in range(reps):
            bar.wait(); f(); done.wait()
    th = [threading.Thread(target=w, args=(f,)) for f in fs]
    for t in th:
        t.start()
    ts = []
    for _ in range(reps):
        bar.wait(); t0 = time.perf_counter(); done.wait(); ts.append(time.perf_counter() - t0)
    for t in th:
        t.join()
    return min(ts[1:])


n = 1 << 20
srs = khip.Srs.create(0, n)
sc = rs(n)
dt = best(lambda: srs.msm(sc))
print(f"kh_msm 2^20 host scalars (PCIe incl.): {1e3 * dt:.3f} ms = {n / dt / 1e6:.1f} Mscalar/s")
srs.close()
n16 = 1 << 16
g = khip.srs_generate(0, 0, n16)
srs16 = khip.Srs(0, g); srs16.set_lagrange(16, g)
cols = rs(15 * n16).reshape(15, n16, 4)
dt = best(lambda: srs16.msm_batch(cols, basis=16))
print(f"kh_msm_batch 15 x 2^16 host scalars: {1e3 * dt:.3f} ms")
x = [np.ascontiguousarray(c) for c in rs(15 * n16).reshape(15, n16, 4)]
xb = np.ascontiguousarray(np.stack(x))
# in place on buffers that exist (ifft_in_place on the caller's Vec): the values change from call to call, the time does not depend on them
dt = best(lambda: [khip.ntt(0, c, 16, True, in_place=True) for c in x])
print(f"interpolate 15 x 2^16, one call per column, ONE thread : {1e3 * dt:.3f} ms  ({15 * n16 * 64 / dt / 1e9:.1f} GB/s both directions)")
dt = best(lambda: khip.ntt(0, xb, 16, True, in_place=True))
print(f"interpolate 15 x 2^16, one batched call               : {1e3 * dt:.3f} ms")
dt = threaded([(lambda c=c: khip.ntt(0, c, 16, True, in_place=True)) for c in x])
print(f"interpolate 15 x 2^16, one call per column, 15 threads : {1e3 * dt:.3f} ms")
dt = best(lambda: [khip.ntt(0, c, 16, True) for c in x])
print(f"  (the same, ONE thread, into a FRESH 2 MB host allocation per call: {1e3 * dt:.3f} ms -- the runtime pins new pageable memory on first use)")
x19 = rs(19 * n16).reshape(19, n16, 4)
dt = best(lambda: khip.ntt(0, x19, 16, True))
print(f"  (round 4's line: kh_ntt 19 x 2^16 batched incl. the harness's numpy copy: {1e3 * dt:.3f} ms; round 4: 10.473)")
y = [np.ascontiguousarray(c).reshape(1, n16, 4) for c in rs(16 * n16).reshape(16, n16, 4)]
yb = np.ascontiguousarray(np.concatenate(y))
ref = [khip.lde(0, c, 16, 3) for c in y]
ob = [np.ones((1, 8 * n16, 4), np.uint64) for _ in range(16)]          # destination buffers that exist already (no first-touch page faults in the timing)
obb = np.ones((16, 8 * n16, 4), np.uint64)
dt = best(lambda: [khip.lde(0, c, 16, 3, out=o) for c, o in zip(y, ob)])
print(f"extend 16 x 2^16 -> 2^19, one call per column, ONE thread : {1e3 * dt:.3f} ms  ({16 * n16 * 288 / dt / 1e9:.1f} GB/s)")
dt = best(lambda: khip.lde(0, yb, 16, 3, out=obb))
print(f"extend 16 x 2^16 -> 2^19, one batched call               : {1e3 * dt:.3f} ms  ({16 * n16 * 288 / dt / 1e9:.1f} GB/s)")
outs = [None] * 16


def ext(i):
    outs[i] = khip.lde(0, y[i], 16, 3, out=ob[i])


dt = threaded([(lambda i=i: ext(i)) for i in range(16)])
print(f"extend 16 x 2^16 -> 2^19, one call per column, 16 threads : {1e3 * dt:.3f} ms  ({16 * n16 * 288 / dt / 1e9:.1f} GB/s)")
assert all(np.array_equal(outs[i], ref[i]) for i in range(16)), "threaded extension differs"
assert np.array_equal(khip.lde(0, yb, 16, 3).reshape(16, -1, 4), np.concatenate(ref).reshape(16, -1, 4)), "batched extension differs"
print("threaded and batched results equal the sequential ones: OK")
