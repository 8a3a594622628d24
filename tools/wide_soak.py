#!/usr/bin/env python3
"""Soak of the wide-window MSM path against the C oracle: random sizes (not powers of two), random scalar distributions (uniform, few distinct values, small,
sparse, one window only, all equal), random offsets, k = 1 / 2, both curves, degenerate bases now and then -- every result bit-exact or the script stops.
Usage: wide_soak.py [seconds]   (run on the GPU box: `gpurun -- python tools/wide_soak.py 60`)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip  # noqa: E402
from oracle import cref  # noqa: E402   (checker)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
khip.init(0)
khip.set_wide_min_n(4096)
rng = np.random.default_rng(int(time.time()) & 0xffff)
print("seed state", rng.bit_generator.state["state"]["state"] & 0xffffffff)


def scalars(kind, n):
    s = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 61) - 1)
    if kind == "few":
        pool = s[:int(rng.integers(1, 9))]
        s = pool[rng.integers(0, len(pool), n)]
    elif kind == "small":
        s[:, 1:] = 0; s[:, 0] &= np.uint64((1 << int(rng.integers(1, 40))) - 1)
    elif kind == "sparse":
        s[rng.random(n) < 0.9] = 0
    elif kind == "window":
        w = int(rng.integers(0, 13)); v = rng.integers(0, 1 << 20, n).astype(object)
        s = cref.ints_to_limbs([int(x) << (20 * w) if 20 * w + 20 <= 253 else int(x) & 0x1fff for x in v]) if True else s
    elif kind == "equal":
        s = np.repeat(s[:1], n, 0)
    return np.ascontiguousarray(s)


t_end = time.time() + budget
runs = 0
while time.time() < t_end:
    cid = int(rng.integers(0, 2))
    n_srs = int(rng.integers(4096, 30000))
    if rng.random() < 0.15:                                 # degenerate bases
        base = khip.srs_generate(cid, 0, 4)
        g = base[rng.integers(0, int(rng.integers(1, 4)), n_srs)]
        srs = khip.Srs(cid, np.ascontiguousarray(g))
    else:
        srs = khip.Srs.create(cid, n_srs)
        g = srs.get_g()
    for _ in range(4):
        # the sort's staging (round 6): the default, and settings that force several passes / straddling buckets / the fall-back at these sizes
        stage, passes = [(28672, 2), (2560, 8), (4096, 3), (0, 2), (3000, 1), (2304, 2)][int(rng.integers(0, 6))]
        khip.set_sort_staging(stage, passes)
        kind = ["uniform", "few", "small", "sparse", "window", "equal"][int(rng.integers(0, 6))]
        off = int(rng.integers(0, n_srs - 4096 + 1))
        n = int(rng.integers(4096, n_srs - off + 1))
        k = int(rng.integers(1, 3))
        mont = bool(rng.integers(0, 2)) if kind in ("uniform", "few", "sparse", "equal") else False
        sc = scalars(kind, n * k)
        d = khip.DevBuf(sc.nbytes).upload(sc)
        got, ginf = srs.msm_batch_dev(d.ptr, n, k, offset=off, mont=mont)
        d.free()
        assert any(nm == "reduce_a1" for nm, _ in khip.last_timings()), "not the wide path"
        for j in range(k):
            want, winf = cref.msm(cid, g[off:off + n], sc[j * n:(j + 1) * n], scalars_mont=mont, threads=16)
            assert bool(ginf[j]) == bool(winf) and (winf or np.array_equal(got[j], want)), (cid, n_srs, kind, off, n, k, j, mont, stage, passes)
        runs += 1
    srs.close()
khip.set_sort_staging()
print(f"wide soak: {runs} MSMs bit-exact")
