"""The coefficient-vector operations around the opening proof (csrc/poly.hip) against their definitions in the
oracle: combine_polys (utils.rs:103-206), b_init (ipa.rs:863-888), evaluate_chunks (chunked_polynomial.rs:21-28),
divide_by_vanishing_poly (prover.rs:903).  combine_polys and b_init are additionally pinned by the reference's
opening-proof bytes in test_gpu_open_kat.py."""
import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def _limbs(F, vals):
    return cref.ints_to_limbs([F.to_mont(v) for v in vals])


def _ints(F, limbs):
    return [F.from_mont(v) for v in cref.limbs_to_ints(limbs)]


def _rand(rnd, F, k):
    return [int.from_bytes(rnd.bytes(40), "little") % F.p for _ in range(k)]


@pytest.mark.parametrize("fid,cid", [(0, 0), (1, 1)])
def test_combine_polys_and_b_init(khip, fid, cid):
    c = P.CURVES[cid]; F = c.scalar
    rnd = np.random.default_rng(31 + fid)
    n = 256
    # lengths: empty, shorter than a chunk, exactly one chunk, several chunks, more blinder chunks than coefficients need
    shapes = [(0, 1), (17, 1), (256, 1), (700, 3), (300, 4), (1, 2)]
    plnms = [(_rand(rnd, F, ln), _rand(rnd, F, ch)) for ln, ch in shapes]
    polyscale = _rand(rnd, F, 1)[0]
    want, _ = P.combine_polys(c, plnms, polyscale, n)
    bufs = []
    for coeffs, _b in plnms:
        d = khip.DevBuf(max(len(coeffs), 1) * 32)
        if coeffs:
            d.upload(_limbs(F, coeffs))
        bufs.append(d)
    out = khip.DevBuf(n * 32)
    plen = khip.combine_polys_dev(fid, bufs, [len(p[0]) for p in plnms], [len(p[1]) for p in plnms], _limbs(F, [polyscale])[0], n, out)
    got = _ints(F, out.download((n, 4)))
    assert plen == 256 and got[:len(want)] == want and not any(got[len(want):])
    # no polynomial at all: the zero polynomial
    assert khip.combine_polys_dev(fid, [], [], [], _limbs(F, [polyscale])[0], n, out) == 0
    assert not out.download((n, 4)).any()
    elm = _rand(rnd, F, 3); evalscale = _rand(rnd, F, 1)[0]
    khip.b_init_dev(fid, _limbs(F, elm), _limbs(F, [evalscale])[0], n, out)
    assert _ints(F, out.download((n, 4))) == P.b_init_vector(F, elm, evalscale, n)
    for d in bufs + [out]:
        d.free()


@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_evaluate_chunks(khip, fid, F):
    rnd = np.random.default_rng(41 + fid)
    for length, chunk, nch in [(300, 128, 4), (128, 128, 1), (1, 64, 1), (70000, 65536, 2), (0, 16, 2), (1000, 1000, 1)]:
        coeffs = _rand(rnd, F, length)
        pts = _rand(rnd, F, 2) + [0, 1]
        d = khip.DevBuf(max(length, 1) * 32)
        if length:
            d.upload(_limbs(F, coeffs))
        got = khip.evaluate_chunks_dev(fid, d, length, chunk, nch, _limbs(F, pts))
        for pi, x in enumerate(pts):
            for ci in range(nch):
                seg = coeffs[ci * chunk:(ci + 1) * chunk]
                acc = 0
                for v in reversed(seg):
                    acc = (acc * x + v) % F.p
                assert F.from_mont(P.from_limbs(got[pi, ci])) == acc, (length, chunk, pi, ci)
        d.free()
    d = khip.DevBuf(32 * 300).upload(_limbs(F, _rand(rnd, F, 300)))
    with pytest.raises(khip.KhError):                      # 300 coefficients do not fit 2 chunks of 128
        khip.evaluate_chunks_dev(fid, d, 300, 128, 2, _limbs(F, [5]))
    d.free()


@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_divide_by_vanishing_poly(khip, fid, F):
    """f = q (x^n - 1) + r with deg r < n; the prover's case is r = 0 (f built as q * Z_H), also checked."""
    rnd = np.random.default_rng(51 + fid)
    logn = 8; n = 1 << logn
    for length in (8 * n - 5, 3 * n, n + 1, n, 7, 0):
        f = _rand(rnd, F, length)
        fd = khip.DevBuf(max(length, 1) * 32)
        if length:
            fd.upload(_limbs(F, f))
        qd = khip.DevBuf(max(length - n, 1) * 32); rd = khip.DevBuf(n * 32)
        khip.divide_by_vanishing_poly_dev(fid, fd, length, logn, qd, rd)
        q = _ints(F, qd.download((max(length - n, 0), 4))) if length > n else []
        r = _ints(F, rd.download((n, 4)))
        # q (x^n - 1) + r == f, coefficient by coefficient
        back = [0] * max(length, n)
        for i, v in enumerate(q):
            back[i + n] = (back[i + n] + v) % F.p
            back[i] = (back[i] - v) % F.p
        for i, v in enumerate(r):
            back[i] = (back[i] + v) % F.p
        assert back[:length] == f and not any(back[length:])
        for d in (fd, qd, rd):
            d.free()
    q0 = _rand(rnd, F, 5 * n)
    f = [0] * (6 * n)
    for i, v in enumerate(q0):
        f[i + n] = (f[i + n] + v) % F.p
        f[i] = (f[i] - v) % F.p
    fd = khip.DevBuf(len(f) * 32).upload(_limbs(F, f)); qd = khip.DevBuf(5 * n * 32); rd = khip.DevBuf(n * 32)
    khip.divide_by_vanishing_poly_dev(fid, fd, len(f), logn, qd, rd)
    assert _ints(F, qd.download((5 * n, 4))) == q0 and not rd.download((n, 4)).any()
    for d in (fd, qd, rd):
        d.free()


def test_poly_lincomb(khip):
    F = P.Fp; rnd = np.random.default_rng(111)
    lens = [100, 37, 0, 100, 1]
    polys = [_rand(rnd, F, ln) for ln in lens]
    sc = _rand(rnd, F, len(lens))
    bufs = [khip.DevBuf(max(ln, 1) * 32) for ln in lens]
    for b, p in zip(bufs, polys):
        if p:
            b.upload(_limbs(F, p))
    out = khip.DevBuf(120 * 32)
    khip.poly_lincomb_dev(0, bufs, lens, _limbs(F, sc), out, 120)
    want = [sum(s * p[i] for s, p in zip(sc, polys) if i < len(p)) % F.p for i in range(120)]
    assert _ints(F, out.download((120, 4))) == want
    with pytest.raises(khip.KhError):
        khip.poly_lincomb_dev(0, bufs, lens, _limbs(F, sc), out, 50)
    for b in bufs + [out]:
        b.free()


def test_evaluate_chunks_batch(khip):
    """All polynomials of a proof in one launch: same values as one call per polynomial (prover.rs:1028-1128)."""
    F = P.Fp; rnd = np.random.default_rng(121)
    shapes = [(500, 4), (128, 1), (0, 1), (300, 3), (1, 1)]
    polys = [_rand(rnd, F, ln) for ln, _ in shapes]
    bufs = []
    for p in polys:
        b = khip.DevBuf(max(len(p), 1) * 32)
        if p:
            b.upload(_limbs(F, p))
        bufs.append(b)
    pts = _rand(rnd, F, 2)
    got = khip.evaluate_chunks_batch_dev(0, bufs, [ln for ln, _ in shapes], [c for _, c in shapes], 128, _limbs(F, pts))
    for j, (b, (ln, c)) in enumerate(zip(bufs, shapes)):
        one = khip.evaluate_chunks_dev(0, b, ln, 128, c, _limbs(F, pts))
        assert np.array_equal(got[j], one)
    for b in bufs:
        b.free()


@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_coset_ntt_is_one_eighth_of_the_lde(khip, fid, F):
    """The multi-GPU decomposition of the d8 extension: coset r = kh_coset_ntt_dev with shift w_{8n}^r, and
    lde[8 i + r] = coset_r[i]; a coset's "next row" is its own next element (z(x w) stays inside the coset)."""
    rnd = np.random.default_rng(131 + fid)
    logn = 9; n = 1 << logn; batch = 3
    coeffs = _limbs(F, _rand(rnd, F, batch * n)).reshape(batch, n, 4)
    want = khip.lde(fid, coeffs, logn, 3)                                  # batch x 8n
    om8 = F.root_of_unity(logn + 3)
    src = khip.DevBuf(coeffs.nbytes).upload(coeffs); dst = khip.DevBuf(coeffs.nbytes)
    for r in range(8):
        khip.coset_ntt_dev(fid, src, logn, _limbs(F, [pow(om8, r, F.p)])[0], dst, batch)
        got = dst.download((batch, n, 4))
        assert np.array_equal(got, want[:, r::8])
    # in place, and the trivial coset is the plain NTT
    khip.coset_ntt_dev(fid, src, logn, _limbs(F, [1])[0], src, batch)
    assert np.array_equal(src.download((batch, n, 4)), khip.ntt(fid, coeffs, logn, False))
    src.free(); dst.free()
