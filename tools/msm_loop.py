#!/usr/bin/env python3
"""A bare loop of 2^log_n-point Vesta MSMs for rocprofv3: msm_loop.py {wide|narrow} {sync|pipe} [steps] [log_n] [depth]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip  # noqa: E402

mode, how = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
log_n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 2
n = 1 << log_n
khip.init(0)
khip.set_wide_min_n(n if mode == "wide" else 0)
srs = khip.Srs.create(khip.VESTA, n)
sc = np.random.default_rng(1234).integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 61) - 1)
d = khip.DevBuf(sc.nbytes).upload(sc)
if how == "sync":
    for _ in range(steps):
        srs.msm_batch_dev(d.ptr, n, 1)
else:
    pending = []
    for _ in range(steps):
        pending.append(srs.msm_submit(d.ptr, n, 1))
        if len(pending) >= depth:
            srs.msm_wait(pending.pop(0))
    while pending:
        srs.msm_wait(pending.pop(0))
khip.sync()
