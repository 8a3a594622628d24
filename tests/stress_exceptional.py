"""Manual GPU stress (not collected by pytest): MSMs over degenerate bases (one point repeated, two points, P / -P pairs) with full, 20-bit and 3-bit scalars
force equal-point doublings, cancellations and identities through every tree and bucket kernel (incl. the lane-cooperative ones); checked against the C oracle."""
import sys, numpy as np
sys.path.insert(0, '.')
from proof_systems_amd import khip
from oracle import cref
khip.init(0)
rng = np.random.default_rng(7)
def rs(k, bits=253):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64)
    if bits <= 64:
        a[:, 1:] = 0; a[:, 0] &= np.uint64((1 << bits) - 1)
    else:
        a[:, 3] &= np.uint64((1 << 61) - 1)
    return a
ok = True
for cid in (0, 1):
    base = khip.srs_generate(cid, 0, 8)
    for n in (1 << 12, 1 << 16):
        for variant in ("all_same", "two_points", "pairs_opposite"):
            g = np.tile(base[0], (n, 1))
            if variant == "two_points":
                g[1::2] = base[1]
            if variant == "pairs_opposite":
                neg = base[0].copy()
                fid = 1 if cid == 0 else 0
                neg[4:] = cref.field_op(fid, "sub", np.zeros((1, 4), np.uint64), base[0, 4:].reshape(1, 4))[0]
                g[1::2] = neg
            srs = khip.Srs(cid, g)
            for bits in (253, 20, 3):
                sc = rs(n, bits)
                for k in (1, 2):
                    scs = np.concatenate([sc, sc[::-1]]) if k == 2 else sc
                    d = khip.DevBuf(scs.nbytes).upload(scs)
                    got, ginf = srs.msm_batch_dev(d.ptr, n, k, mont=False)
                    for j in range(k):
                        w, winf = cref.msm(cid, g, scs[j * n:(j + 1) * n], scalars_mont=False, threads=8)
                        good = bool(ginf[j]) == bool(winf) and (winf or np.array_equal(got[j], w))
                        ok &= good
                        if not good: print("MISMATCH", cid, n, variant, bits, k, j)
                    d.free()
            srs.close()
print("stress ok:", ok)
