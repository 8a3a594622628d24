#!/usr/bin/env python3
"""The 2^20 MSM from HOST scalars (pageable memory, what SRS::commit_non_hiding(&DensePolynomial) hands over, poly-commitment/src/ipa.rs:638-683):
one at a time through kh_msm (two half-range jobs with chunked uploads), and pipelined through kh_msm_submit_host / kh_msm_wait with 2 / 3 in flight,
beside the device-resident pipeline of bench.py.  Distinct host buffers per MSM in flight (a caller's polynomials are different Vecs); results checked
against each other.  KH_HOST_SPLIT_MIN=0 / KH_HOST_CHUNK_MIN=0 in the environment give the round-5 behaviour for an A/B on the same box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proof_systems_amd.khip as khip  # noqa: E402

khip.init(0)
rng = np.random.default_rng(3)
logn = int(os.environ.get("LOGN", "20"))
n = 1 << logn
steps = int(os.environ.get("STEPS", "20"))


def rs(m):
    s = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 61) - 1)
    return s


srs = khip.Srs.create(0, n)
bufs = [rs(n) for _ in range(4)]
ref = [srs.msm(b) for b in bufs]


def single():
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); r = srs.msm(bufs[i % 4]); ts.append(time.perf_counter() - t0)
        assert np.array_equal(r[0], ref[i % 4][0])
    return ts


ts = single()
print(f"kh_msm 2^{logn} from pageable host scalars, one at a time: min {1e3 * min(ts[1:]):.3f} ms  median {1e3 * sorted(ts[1:])[len(ts[1:]) // 2]:.3f} ms = {n / sorted(ts[1:])[len(ts[1:]) // 2] / 1e6:.0f} Mscalar/s")


def pipelined(depth, host=True, dev=None):
    vals = []
    for region in range(3):
        q = []
        t0 = time.perf_counter()
        for i in range(steps):
            if len(q) == depth:
                j, t = q.pop(0)
                out, inf = khip.Srs.msm_wait(t)
                assert np.array_equal(out[0], ref[j][0])
            j = i % 4
            q.append((j, srs.msm_submit_host(bufs[j]) if host else srs.msm_submit(dev[j].ptr, n, 1)))
        while q:
            j, t = q.pop(0)
            out, inf = khip.Srs.msm_wait(t)
            assert np.array_equal(out[0], ref[j][0])
        dt = time.perf_counter() - t0
        vals.append(n * steps / dt / 1e6)
    return vals


for depth in (1, 2, 3):
    v = pipelined(depth)
    print(f"kh_msm_submit_host, {depth} in flight, {steps} steps x 3 regions: {' / '.join(f'{x:.0f}' for x in v)} Mscalar/s (median {sorted(v)[1]:.0f}; {1e3 * n / sorted(v)[1] / 1e6:.3f} ms per MSM)")
dev = [khip.DevBuf(n * 32).upload(b) for b in bufs]
v = pipelined(2, host=False, dev=dev)
print(f"kh_msm_submit (device-resident scalars), 2 in flight: {' / '.join(f'{x:.0f}' for x in v)} Mscalar/s (median {sorted(v)[1]:.0f})")
srs.close()
