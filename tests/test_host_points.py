"""The library's host-side group operations (kh_points_sum: the fold of per-GPU partial sums; kh_points_add: the second half of a masking whose
blinding points were computed while the device was busy, ipa.rs:605-622) against the oracle's C port.  Host code: runs without a GPU."""
import numpy as np
import pytest

from oracle import cref


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    return k


def _neg(cid, p):
    q = p.copy()
    q[4:] = cref.field_op(1 if cid == 0 else 0, "sub", np.zeros((1, 4), np.uint64), p[4:].reshape(1, 4))[0]
    return q


@pytest.mark.parametrize("cid", [0, 1])
def test_points_add_pairwise(khip, cid):
    g = cref.srs_generate(cid, 0, 40)
    rng = np.random.default_rng(11 + cid)
    a = g[rng.integers(0, 40, size=64)]
    b = g[rng.integers(0, 40, size=64)]
    ai = (rng.integers(0, 8, size=64) == 0).astype(np.uint8)
    bi = (rng.integers(0, 8, size=64) == 0).astype(np.uint8)
    a[0], b[0], ai[0], bi[0] = g[3], g[3], 0, 0                      # doubling
    a[1], b[1], ai[1], bi[1] = g[4], _neg(cid, g[4]), 0, 0           # P + (-P)
    ai[2] = bi[2] = 1                                                # infinity + infinity
    got, gi = khip.points_add(cid, a, ai, b, bi)
    for j in range(64):
        if ai[j] and bi[j]:
            want, winf = None, True
        elif ai[j]:
            want, winf = b[j], False
        elif bi[j]:
            want, winf = a[j], False
        else:
            want, winf = cref.point_add(cid, a[j], b[j], False, False)
        assert bool(gi[j]) == bool(winf), j
        if not winf:
            assert np.array_equal(got[j], want), j
        else:
            assert not got[j].any()
    assert gi[1] and gi[2] and not gi[0]
    # no flags at all = no point at infinity; an empty list is fine
    got2, gi2 = khip.points_add(cid, a[3:10], None, b[3:10], None)
    for j in range(7):
        want, winf = cref.point_add(cid, a[3 + j], b[3 + j], False, False)
        assert bool(gi2[j]) == bool(winf) and (winf or np.array_equal(got2[j], want))
    e, ei = khip.points_add(cid, np.zeros((0, 8), np.uint64), None, np.zeros((0, 8), np.uint64), None)
    assert e.shape == (0, 8) and ei.shape == (0,)


@pytest.mark.parametrize("cid", [0, 1])
def test_points_sum_fold(khip, cid):
    g = cref.srs_generate(cid, 0, 9)
    pts = np.concatenate([g, g[:1], _neg(cid, g[1])[None]])
    inf = np.zeros(len(pts), np.uint8); inf[3] = 1
    got, ginf = khip.points_sum(cid, pts, inf)
    acc, ainf = pts[0].copy(), False
    for i in range(1, len(pts)):
        if not inf[i]:
            acc, ainf = cref.point_add(cid, acc, pts[i], ainf, False)
    assert ginf == ainf and np.array_equal(got, acc)
    assert khip.points_sum(cid, np.stack([g[1], _neg(cid, g[1])]))[1]
    assert khip.points_sum(cid, np.zeros((0, 8), np.uint64))[1]
