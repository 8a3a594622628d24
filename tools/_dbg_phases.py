import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import proof_systems_amd.khip as khip
n = 1 << 20
khip.init(0)
srs = khip.Srs.create(khip.VESTA, n)
sc = np.random.default_rng(1).integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 61) - 1)
d = khip.DevBuf(sc.nbytes).upload(sc)
ref = None
for og in (1, 4, 64, 2048):
    os.environ["KH_WIDE_OG"] = str(og)
    acc = {}
    for _ in range(6):
        out = srs.msm_batch_dev(d.ptr, n, 1)
        for k, v in khip.last_timings(): acc.setdefault(k, []).append(v)
    if ref is None: ref = out
    assert np.array_equal(ref[0], out[0])
    print("og", og, "  ".join(f"{k} {np.median(v)*1e3:.0f}" for k, v in acc.items()))
