"""Parity tests proper: the HIP path (through the C ABI of include/kimchi_hip.h) against the
CPU oracle on the same seeded inputs, bit-exact.  Run on a real MI355X: `pytest -m gpu`."""
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


def rand_fe(rng, n, F):
    """n uniformly random field elements as (n,4) limb arrays (any value < p is a valid
    Montgomery representation, like ark-ff's Fp::rand)."""
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    while filled < n:
        m = n - filled
        c = rng.integers(0, 1 << 64, size=(2 * m + 8, 4), dtype=np.uint64)
        c[:, 3] &= np.uint64((1 << 63) - 1)
        vals = [P.from_limbs(r) for r in c]
        ok = [i for i, v in enumerate(vals) if v < F.p][:m]
        out[filled:filled + len(ok)] = c[ok]
        filled += len(ok)
    return out


def rand_fe_fast(rng, n):
    """Uniform below 2^253 (< p): fast path for big inputs."""
    c = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    c[:, 3] &= np.uint64((1 << 61) - 1)
    return c


def edge_fe(F):
    vals = [0, 1, 2, F.p - 1, F.p - 2, (F.p - 1) // 2, 1 << 254, F.R, F.R2, (1 << 32) - 1, 1 << 32, (1 << 64) - 1,
            1 << 64, (1 << 128) - 1, (1 << 192) + 5, F.p - (1 << 32), F.p - (1 << 200)]
    return cref.ints_to_limbs([v % F.p for v in vals])


# ------------------------------------------------------------------ field arithmetic on the device
@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_field_ops(khip, fid, F):
    rng = np.random.default_rng(100 + fid)
    e = edge_fe(F)
    a = np.concatenate([np.repeat(e, len(e), axis=0), rand_fe(rng, 20000, F)])
    b = np.concatenate([np.tile(e, (len(e), 1)), rand_fe(rng, 20000, F)])
    for op in ("mul", "add", "sub"):
        got = khip.debug_field_op(fid, op, a, b)
        want = cref.field_op(fid, op, a, b)
        assert np.array_equal(got, want), op
    for op in ("to_mont", "from_mont", "sqr"):
        got = khip.debug_field_op(fid, op, a)
        want = cref.field_op(fid, op, a)
        assert np.array_equal(got, want), op
    got = khip.debug_field_op(fid, "neg", a)
    want = cref.field_op(fid, "sub", np.zeros_like(a), a)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_field29_ops(khip, fid, F):
    """field29.cuh (nine 29-bit limbs, R' = 2^261, lazy reduction -- the arithmetic of k_accumulate29): product, squaring,
    the repacking, and the unreduced 32 X form of a table coordinate, bit-exact against the oracle after conversion back."""
    rng = np.random.default_rng(300 + fid)
    e = edge_fe(F)
    a = np.concatenate([np.repeat(e, len(e), axis=0), rand_fe(rng, 30000, F)])
    b = np.concatenate([np.tile(e, (len(e), 1)), rand_fe(rng, 30000, F)])
    want = cref.field_op(fid, "mul", a, b)
    assert np.array_equal(khip.debug_field_op(fid, "mul29", a, b), want)
    assert np.array_equal(khip.debug_field_op(fid, "mul29_32x", a, b), want)
    assert np.array_equal(khip.debug_field_op(fid, "sqr29", a), cref.field_op(fid, "sqr", a))
    assert np.array_equal(khip.debug_field_op(fid, "pack29", a), a)


@pytest.mark.parametrize("cid", [0, 1])
def test_madd29(khip, cid):
    """The lazy mixed addition of field29.cuh: 2P + Q equals the oracle's group law on generic operands, and every
    exceptional case (Q = 2P, Q = -2P, identity accumulator) is DECLINED (flag 2) instead of computed -- the accumulation
    kernel hands such tasks to the exact path."""
    n = 2048
    g = cref.srs_generate(cid, 0, n, threads=8)
    rng = np.random.default_rng(25 + cid)
    p, q = g, g[rng.permutation(n)].copy()
    base_fid = 1 if cid == 0 else 0
    pinf = np.zeros(n, np.uint8); qinf = np.zeros(n, np.uint8)
    q[0] = p[0]                                                           # 2P + P: generic
    for i in (5, 70):                                                     # Q = 2P
        q[i] = cref.point_add(cid, p[i], p[i], False, False)[0]
    for i in (6, 71):                                                     # Q = -2P
        d = cref.point_add(cid, p[i], p[i], False, False)[0]
        d[4:] = cref.field_op(base_fid, "sub", np.zeros((1, 4), np.uint64), d[4:].reshape(1, 4))[0]
        q[i] = d
    pinf[[7, 72]] = 1                                                     # accumulator = identity
    got, flag = khip.debug_point_op(cid, 6, p, q, pinf, qinf)
    for i in range(n):
        if i in (5, 70, 6, 71, 7, 72):
            assert flag[i] == 2, i
            continue
        d, _ = cref.point_add(cid, p[i], p[i], False, False)
        w, winf = cref.point_add(cid, d, q[i], False, False)
        assert flag[i] == 0 and not winf and np.array_equal(got[i], w), i


# ------------------------------------------------------------------ group law on the device
@pytest.mark.parametrize("cid", [0, 1])
def test_point_ops(khip, cid):
    c = P.CURVES[cid]
    n = 512
    g = cref.srs_generate(cid, 0, n, threads=8)
    rng = np.random.default_rng(5 + cid)
    perm = rng.permutation(n)
    p, q = g, g[perm].copy()
    # force exceptional cases: q == p, q == -p, infinity operands
    q[0] = p[0]
    q[1] = p[1]
    negy = cref.field_op(1 - cid if cid == 0 else 0, "sub", np.zeros((1, 4), np.uint64), p[1, 4:].reshape(1, 4))
    q[1, 4:] = negy[0]
    pinf = np.zeros(n, np.uint8); qinf = np.zeros(n, np.uint8)
    pinf[2] = 1; qinf[3] = 1; pinf[4] = 1; qinf[4] = 1
    for op in (0, 2):
        got, ginf = khip.debug_point_op(cid, op, p, q, pinf, qinf)
        for i in range(n):
            w, winf = cref.point_add(cid, p[i], q[i], bool(pinf[i]), bool(qinf[i]))
            assert bool(ginf[i]) == winf, (op, i)
            if not winf:
                assert np.array_equal(got[i], w), (op, i)
    got, ginf = khip.debug_point_op(cid, 1, p, q, pinf, qinf)
    for i in range(0, n, 7):
        w, winf = cref.point_add(cid, p[i], p[i], bool(pinf[i]), bool(pinf[i]))
        assert bool(ginf[i]) == winf and (winf or np.array_equal(got[i], w))
    # 2P - Q through dbl + mixed add with a non-trivial ZZ (includes 2P - P... = P when q == p)
    got, ginf = khip.debug_point_op(cid, 3, p, q, pinf, qinf)
    base_fid = 1 if cid == 0 else 0
    for i in range(0, n, 3):
        d, dinf = cref.point_add(cid, p[i], p[i], bool(pinf[i]), bool(pinf[i]))
        if qinf[i]:
            w, winf = d, dinf
        else:
            nq = q[i].copy()
            nq[4:] = cref.field_op(base_fid, "sub", np.zeros((1, 4), np.uint64), q[i, 4:].reshape(1, 4))[0]
            w, winf = cref.point_add(cid, d, nq, dinf, False)
        assert bool(ginf[i]) == winf and (winf or np.array_equal(got[i], w)), i


@pytest.mark.parametrize("cid", [0, 1])
def test_lane_cooperative_addition(khip, cid):
    """coop.cuh: the XYZZ addition spread over four lanes (used by the tree phases of the bucket reduction) equals the
    oracle's group law on generic, equal (doubling fallback), opposite and identity operands, with trivial (op 4) and
    non-trivial (op 5: both operands doubled first) ZZ / ZZZ; 517 pairs leave a ragged last wave."""
    n = 517
    g = cref.srs_generate(cid, 0, n, threads=8)
    rng = np.random.default_rng(15 + cid)
    p, q = g, g[rng.permutation(n)].copy()
    base_fid = 1 if cid == 0 else 0
    for i in (0, 7, 64, 65, 300, 516):
        q[i] = p[i]                                        # equal points
    for i in (1, 66, 301, 515):
        q[i] = p[i]
        q[i, 4:] = cref.field_op(base_fid, "sub", np.zeros((1, 4), np.uint64), p[i, 4:].reshape(1, 4))[0]   # opposite points
    pinf = np.zeros(n, np.uint8); qinf = np.zeros(n, np.uint8)
    pinf[[2, 67, 302]] = 1; qinf[[3, 68, 303]] = 1; pinf[[4, 69]] = 1; qinf[[4, 69]] = 1
    got, ginf = khip.debug_point_op(cid, 4, p, q, pinf, qinf)
    for i in range(n):
        w, winf = cref.point_add(cid, p[i], q[i], bool(pinf[i]), bool(qinf[i]))
        assert bool(ginf[i]) == winf and (winf or np.array_equal(got[i], w)), ("op4", i)
    got, ginf = khip.debug_point_op(cid, 5, p, q, pinf, qinf)
    for i in range(n):
        a, ai = cref.point_add(cid, p[i], p[i], bool(pinf[i]), bool(pinf[i]))
        b, bi = cref.point_add(cid, q[i], q[i], bool(qinf[i]), bool(qinf[i]))
        w, winf = cref.point_add(cid, a, b, ai, bi)
        assert bool(ginf[i]) == winf and (winf or np.array_equal(got[i], w)), ("op5", i)


# ------------------------------------------------------------------ MSM
def _check_msm(khip, cid, g, sc, mont=True, threads=8):
    srs = khip.Srs(cid, g)
    got, ginf = srs.msm(sc, mont=mont)
    want, winf = cref.msm(cid, g, sc, scalars_mont=mont, threads=threads)
    srs.close()
    assert ginf == winf
    if not winf:
        assert np.array_equal(got, want)


def test_msm_kat(khip, golden):
    """kimchi/src/proof.rs:1160-1204 through the device path."""
    c = P.VESTA
    kat = golden["msm_kat"]
    coeffs = [1, 7, 5, 35, 3, 21, 15, 105, 2, 14, 10, 70, 6, 42, 30, 210]
    basis = [c.mul(c.gen, i) for i in range(1, 17)]
    xy = np.zeros((16, 8), np.uint64)
    for i, (x, y) in enumerate(basis):
        xy[i, :4] = P.to_limbs(c.base.to_mont(x)); xy[i, 4:] = P.to_limbs(c.base.to_mont(y))
    for mont in (True, False):
        sc = cref.ints_to_limbs([P.Fp.to_mont(v) if mont else v for v in coeffs])
        got, inf = khip.msm_points(0, xy, sc, mont=mont)
        assert not inf
        assert c.base.from_mont(P.from_limbs(got[:4])) == int(kat["expected_x"])
        assert c.base.from_mont(P.from_limbs(got[4:])) == int(kat["expected_y"])


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("logn", [0, 1, 5, 10, 13])
def test_msm_uniform(khip, cid, logn):
    n = 1 << logn
    F = P.CURVES[cid].scalar
    rng = np.random.default_rng(1000 * cid + logn)
    g = cref.srs_generate(cid, 0, n, threads=8)
    _check_msm(khip, cid, g, rand_fe(rng, n, F))
    _check_msm(khip, cid, g, cref.field_op(F is P.Fq and 1 or 0, "from_mont", rand_fe(rng, n, F)), mont=False)


@pytest.mark.parametrize("cid", [0, 1])
def test_msm_edge_distributions(khip, cid):
    """(b) bench-circuit witness (n-10 ones, 7 zeros, 3 random), (c) all zero,
    (d) repeated points / P and -P in one bucket; plus -1 (every window max), tiny scalars, ragged n."""
    c = P.CURVES[cid]; F = c.scalar
    fid = 0 if F is P.Fp else 1
    n = 3000
    rng = np.random.default_rng(42 + cid)
    g = cref.srs_generate(cid, 0, n, threads=8)
    one = cref.ints_to_limbs([F.R])
    zero = np.zeros((1, 4), np.uint64)
    cases = {
        "bench_circuit": np.concatenate([np.repeat(one, n - 10, 0), np.repeat(zero, 7, 0), rand_fe(rng, 3, F)]),
        "zeros": np.zeros((n, 4), np.uint64),
        "minus_one": np.repeat(cref.ints_to_limbs([F.to_mont(F.p - 1)]), n, 0),
        "small": cref.field_op(fid, "to_mont", cref.ints_to_limbs([int(v) for v in rng.integers(0, 1 << 16, n)])),
        "same_scalar": np.repeat(rand_fe(rng, 1, F), n, 0),
        "half_boundary": cref.field_op(fid, "to_mont", cref.ints_to_limbs([(1 << (11 * (i % 20))) * ((1 << 10) + (i & 1)) % F.p for i in range(n)])),
    }
    for name, sc in cases.items():
        _check_msm(khip, cid, g, sc)
    # repeated points incl. P and -P with equal scalars, and infinity inputs, through kh_msm_points
    base_fid = 1 - fid
    pts = np.concatenate([g[:5], g[:5], g[:5]])
    pts[10:15, 4:] = cref.field_op(base_fid, "sub", np.zeros((5, 4), np.uint64), g[:5, 4:])
    sc = np.concatenate([rand_fe(rng, 5, F)] * 3)
    inf = np.zeros(15, np.uint8); inf[7] = 1
    got, ginf = khip.msm_points(cid, pts, sc, inf=inf)
    want, winf = cref.msm(cid, pts, sc, inf=inf)
    assert ginf == winf and (winf or np.array_equal(got, want))
    got, ginf = khip.msm_points(cid, pts[[0, 10]], sc[[0, 10]])        # P*s + (-P)*s = infinity
    assert ginf
    # empty input -> infinity
    got, ginf = khip.msm_points(cid, np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64))
    assert ginf
    # ragged: fewer scalars than bases, offset into the basis (commit_non_hiding's last chunk, ipa.rs:663-676)
    srs = khip.Srs(cid, g)
    sc = rand_fe(rng, 777, F)
    got, ginf = srs.msm(sc, offset=100)
    want, winf = cref.msm(cid, g[100:877], sc)
    assert ginf == winf and np.array_equal(got, want)
    got, ginf = srs.msm(rand_fe(rng, 50, F), offset=n - 20)            # more scalars than remaining bases
    srs.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_msm_batch_and_lagrange(khip, cid):
    """15 witness-column commits against a Lagrange basis (prover.rs:329-351, ipa.rs:706-728)."""
    c = P.CURVES[cid]; F = c.scalar
    logn = 6; n = 1 << logn
    rng = np.random.default_rng(9 + cid)
    g = cref.srs_generate(cid, 0, n, threads=8)
    bxy, binf = cref.lagrange_basis(cid, g, logn)
    srs = khip.Srs(cid, g)
    srs.set_lagrange(logn, bxy, binf)
    cols = np.stack([rand_fe(rng, n, F) for _ in range(15)])
    cols[3] = 0
    cols[4, : n - 10] = cref.ints_to_limbs([F.R])[0]
    got, ginf = srs.msm_batch(cols, basis=logn)
    for j in range(15):
        want, winf = cref.msm(cid, bxy, cols[j], inf=binf)
        assert bool(ginf[j]) == winf and (winf or np.array_equal(got[j], want)), j
    # commit_evaluations(e) == commit(interpolate(e))  (poly-commitment/tests/ipa_commitment.rs:26-52)
    fid = 0 if F is P.Fp else 1
    coeffs = cref.ntt(fid, cols[0], logn, True)[0]
    want, winf = srs.msm(coeffs)
    assert np.array_equal(got[0], want) and not winf
    srs.close()


# ------------------------------------------------------------------ NTT
@pytest.mark.parametrize("fid", [0, 1])
@pytest.mark.parametrize("logn", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19])
def test_ntt_sizes(khip, fid, logn):
    """Every size from 2^0 to 2^19 (2^20 .. 2^22: test_ntt_large_sizes), both fields, both directions, directly against the oracle's DFT
    ("bit-exact vs oracle for log n in [3, 22], both fields")."""
    rng = np.random.default_rng(31 * logn + fid)
    n = 1 << logn
    batch = 3 if logn <= 13 else 2
    x = rand_fe_fast(rng, batch * n).reshape(batch, n, 4)
    for inverse in (False, True):
        got = khip.ntt(fid, x, logn, inverse)
        want = cref.ntt(fid, x, logn, inverse, threads=8)
        assert np.array_equal(got, want), (logn, inverse)
    # round trip
    assert np.array_equal(khip.ntt(fid, khip.ntt(fid, x, logn, False), logn, True), x)


@pytest.mark.parametrize("fid", [0, 1])
@pytest.mark.parametrize("logn,logb", [(1, 3), (3, 3), (6, 3), (8, 2), (10, 3), (12, 3), (13, 1), (14, 2), (15, 3), (16, 2), (16, 3), (17, 3), (18, 3), (19, 1)])
def test_lde(khip, fid, logn, logb):
    rng = np.random.default_rng(77 * logn + logb + fid)
    n = 1 << logn
    batch = 2
    c = rand_fe_fast(rng, batch * n).reshape(batch, n, 4)
    got = khip.lde(fid, c, logn, logb)
    want = cref.lde(fid, c, logn, logb, threads=8)
    assert np.array_equal(got, want)
    # kimchi/tests/test_domain.rs:25-71: the d1 points are every 2^b-th d8 point
    assert np.array_equal(got[:, :: 1 << logb], cref.ntt(fid, c, logn, False, threads=8))


def test_smoke_entry(khip):
    import __graft_entry__ as ge
    ge.smoke()


@pytest.mark.parametrize("cid", [0, 1])
def test_msm_points_batch(khip, cid):
    """Independent bases per MSM in one pass (the L / R pair of an IPA round, ipa.rs:943-961)."""
    F = P.CURVES[cid].scalar
    rng = np.random.default_rng(321 + cid)
    for n, k in [(1, 2), (34, 2), (515, 3), (2050, 2)]:
        g = cref.srs_generate(cid, 7, n * k, threads=8).reshape(k, n, 8)
        sc = np.stack([rand_fe(rng, n, F) for _ in range(k)])
        inf = np.zeros((k, n), np.uint8)
        if n > 2:
            inf[1, 2] = 1
        got, ginf = khip.msm_points_batch(cid, g, sc, inf=inf)
        for j in range(k):
            want, winf = cref.msm(cid, g[j], sc[j], inf=inf[j])
            assert bool(ginf[j]) == winf and (winf or np.array_equal(got[j], want)), (n, k, j)


def test_msm_submit_wait_pipeline(khip):
    """KH_MSM_SLOTS jobs in flight give the same results as the synchronous calls; one more
    un-waited submit is refused; tickets cannot be waited twice."""
    rng = np.random.default_rng(77)
    n = 1 << 12
    slots = khip.MSM_SLOTS
    g = cref.srs_generate(0, 0, n, threads=8)
    srs = khip.Srs(0, g)
    scs = [rand_fe_fast(rng, n) for _ in range(2 * slots + 1)]
    bufs = [khip.DevBuf(s.nbytes).upload(s) for s in scs]
    want = [srs.msm(s) for s in scs]
    tickets = [srs.msm_submit(bufs[i].ptr, n, 1) for i in range(slots)]
    with pytest.raises(khip.KhError):
        srs.msm_submit(bufs[slots].ptr, n, 1)
    got = []
    for i in range(slots, len(scs)):
        got.append(srs.msm_wait(tickets.pop(0)))
        tickets.append(srs.msm_submit(bufs[i].ptr, n, 1))
    last = tickets[-1]
    got += [srs.msm_wait(t) for t in tickets]
    for (o, i), (wo, wi) in zip(got, want):
        assert bool(i[0]) == wi and np.array_equal(o[0], wo)
    with pytest.raises(khip.KhError):
        srs.msm_wait(last)
    for b in bufs:
        b.free()
    srs.close()


def test_device_producer_then_msm_on_another_slot(khip):
    """An MSM whose device-resident scalars are still being produced on the library's main stream (kh_ntt_dev is
    asynchronous) must wait for them even when it runs on another pipeline slot's stream: slot 0 (= the main stream) is
    kept busy by an un-waited submit, the iNTT queues behind it, the commitment of its output lands on slot 1."""
    rng = np.random.default_rng(99)
    logn = 16; n = 1 << logn
    srs = khip.Srs.create(0, n)
    big = khip.DevBuf(n * 32).upload(rand_fe_fast(rng, n))
    x = rand_fe_fast(rng, n)
    want_coeffs = khip.ntt(0, x, logn, True)[0]
    want, winf = srs.msm(want_coeffs)
    for _ in range(3):
        col = khip.DevBuf(n * 32).upload(x)
        ticket = srs.msm_submit(big.ptr, n, 1)            # occupies slot 0 and its stream
        khip.ntt_dev(0, col, logn, True, 1)               # queued on the same stream, returns immediately
        got, ginf = srs.msm_batch_dev(col.ptr, n, 1)      # slot 1: must see the iNTT output
        srs.msm_wait(ticket)
        assert bool(ginf[0]) == bool(winf) and np.array_equal(got[0], want)
        col.free()
    big.free()
    srs.close()


def test_concurrent_callers(khip):
    """The SRS is Sync + Send and is called from 15 rayon workers at once (kimchi/src/prover.rs:329-351):
    concurrent kh_msm / kh_ntt calls from several host threads give the single-threaded results."""
    import threading
    rng = np.random.default_rng(2718)
    n = 1 << 11
    g = cref.srs_generate(0, 0, n, threads=8)
    srs = khip.Srs(0, g)
    cols = [rand_fe_fast(rng, n) for _ in range(8)]
    want = [srs.msm(c) for c in cols]
    want_ntt = [khip.ntt(0, c, 11, True) for c in cols]
    got = [None] * 8
    got_ntt = [None] * 8

    def work(j):
        got[j] = srs.msm(cols[j])
        got_ntt[j] = khip.ntt(0, cols[j], 11, True)

    th = [threading.Thread(target=work, args=(j,)) for j in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for j in range(8):
        assert got[j][1] == want[j][1] and np.array_equal(got[j][0], want[j][0])
        assert np.array_equal(got_ntt[j], want_ntt[j])
    srs.close()


@pytest.mark.timeout(300)
def test_four_streams_of_small_msms_share_the_chip(khip):
    """The one-launch sort of small jobs (k_sort_fused) keeps 64 blocks spinning on grid barriers; four host threads -- one per
    pipeline slot -- run 40 single and paired commitments each over a 2^14-point basis at the same time, with a few 2^16-sized
    jobs (the large path) mixed in: nobody deadlocks, and every result is the single-threaded one."""
    import threading
    rng = np.random.default_rng(4242)
    logn = 14; n = 1 << logn
    srs = khip.Srs.create(khip.VESTA, 1 << 16)
    cols = [rand_fe_fast(rng, n) for _ in range(6)]
    pair = np.stack(cols[:2])
    big = rand_fe_fast(rng, 1 << 16)
    want = [srs.msm(c) for c in cols]
    want_pair = srs.msm_batch(pair)
    want_big = srs.msm(big)
    errors = []

    def work(t):
        try:
            for it in range(40):
                j = (t + it) % 6
                xy, inf = srs.msm(cols[j])
                assert inf == want[j][1] and np.array_equal(xy, want[j][0])
                if it % 5 == t % 5:
                    bxy, binf = srs.msm_batch(pair)
                    assert np.array_equal(bxy, want_pair[0]) and np.array_equal(binf, want_pair[1])
                if it % 13 == 0:
                    xy, inf = srs.msm(big)
                    assert inf == want_big[1] and np.array_equal(xy, want_big[0])
        except Exception as e:                                               # noqa: BLE001 -- reported below, from the main thread
            errors.append((t, repr(e)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    srs.close()


def test_fifteen_commit_callers_are_coalesced(khip):
    """The reference's witness commitments: 15 threads, each SRS::commit_evaluations_non_hiding on ITS OWN host column over
    the same Lagrange basis, at 2^16.  The library merges callers that arrive in a burst into batched launches; every
    caller must still get exactly its own commitment (checked against one batched call and, for three columns, the oracle),
    and the burst must not cost much more than the batched call (the gate of VERDICT round 1: <= 1.5x, asserted loosely at
    3x because thread start-up jitter on a shared box is part of the number)."""
    import threading
    import time
    rng = np.random.default_rng(99)
    logn = 16; n = 1 << logn
    srs = khip.Srs.create(khip.VESTA, n)
    srs.compute_lagrange(logn)
    cols = np.stack([rand_fe_fast(rng, n) for _ in range(15)])
    cols[3] = 0
    cols[4, : n - 10] = cref.ints_to_limbs([P.Fp.R])[0]                    # the bench circuit's column
    batch = srs.msm_batch(cols, basis=logn)
    t0 = time.perf_counter(); srs.msm_batch(cols, basis=logn); t_batch = time.perf_counter() - t0
    got = [None] * 15
    bar = threading.Barrier(16); done = threading.Barrier(16)

    def work(j):                                                           # a pool that exists before the burst, like rayon's
        for _ in range(4):
            bar.wait()
            got[j] = srs.commit_evaluations_non_hiding(logn, cols[j])
            done.wait()

    th = [threading.Thread(target=work, args=(j,)) for j in range(15)]
    for t in th:
        t.start()
    best = None
    for _ in range(4):
        bar.wait()
        t0 = time.perf_counter()
        done.wait()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        for j in range(15):
            assert bool(got[j][1][0]) == bool(batch[1][j]) and np.array_equal(got[j][0][0], batch[0][j]), j
    for t in th:
        t.join()
    bxy, binf = srs.get_lagrange(logn)
    for j in (0, 4, 14):
        want, winf = cref.msm(0, bxy, cols[j], inf=binf, threads=16)
        assert bool(batch[1][j]) == winf and (winf or np.array_equal(batch[0][j], want))
    assert best < 3 * t_batch + 2e-3, (best, t_batch)
    srs.close()


def test_pallas_vesta_pair_resident(khip):
    """BASELINE config 5: the recursion pair -- a Vesta SRS (coords Fq, scalars Fp) and a Pallas SRS
    (coords Fp, scalars Fq) resident in the same process, their MSMs and both fields' NTTs interleaved:
    all four instantiations MSM<Fq>, MSM<Fp>, NTT<Fp>, NTT<Fq> of one binary, no cross-talk."""
    rng = np.random.default_rng(55)
    n = 1 << 12
    gv = cref.srs_generate(0, 0, n, threads=8)
    gp = cref.srs_generate(1, 0, n, threads=8)
    sv, sp = khip.Srs(0, gv), khip.Srs(1, gp)
    for rep in range(3):
        a, b = rand_fe_fast(rng, n), rand_fe_fast(rng, n)
        tv = sv.msm_submit(khip.DevBuf(a.nbytes).upload(a).ptr, n, 1)
        tp = sp.msm_submit(khip.DevBuf(b.nbytes).upload(b).ptr, n, 1)
        ev = khip.ntt(0, a, 12, True)                     # Fp transform while both MSMs are in flight
        eq = khip.ntt(1, b, 12, True)                     # Fq transform
        (ov, iv), (op, ip_) = sv.msm_wait(tv), sp.msm_wait(tp)
        wv, wiv = cref.msm(0, gv, a, threads=8)
        wp, wip = cref.msm(1, gp, b, threads=8)
        assert bool(iv[0]) == wiv and np.array_equal(ov[0], wv)
        assert bool(ip_[0]) == wip and np.array_equal(op[0], wp)
        assert np.array_equal(ev, cref.ntt(0, a, 12, True)) and np.array_equal(eq, cref.ntt(1, b, 12, True))
    sv.close(); sp.close()


def test_msm_randomized_differential(khip):
    """Seeded differential sweep: random lengths (both the plain and the window-table path), random
    mixes of scalar classes (uniform, zero, one, minus one, small, repeated), random infinity flags,
    both curves, single and batched calls -- each compared bit-for-bit with the oracle."""
    rng = np.random.default_rng(20260925)
    for case in range(36):
        cid = case & 1
        F = P.CURVES[cid].scalar
        fid = 0 if F is P.Fp else 1
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 1000, 1023, 1024, 1025, 2500, 4097]))
        g = cref.srs_generate(cid, int(rng.integers(0, 1000)), n, threads=8)
        cls = rng.integers(0, 6, size=n)
        sc = rand_fe_fast(rng, n)
        one = cref.ints_to_limbs([F.R])[0]
        m1 = cref.ints_to_limbs([F.to_mont(F.p - 1)])[0]
        small = cref.field_op(fid, "to_mont", cref.ints_to_limbs([int(v) for v in rng.integers(0, 70000, n)]))
        sc[cls == 1] = 0
        sc[cls == 2] = one
        sc[cls == 3] = m1
        sc[cls == 4] = small[cls == 4]
        sc[cls == 5] = sc[0]
        if case % 3 == 0:                                     # ad-hoc bases with infinity flags
            inf = (rng.random(n) < 0.1).astype(np.uint8)
            got, ginf = khip.msm_points(cid, g, sc, inf=inf)
            want, winf = cref.msm(cid, g, sc, inf=inf, threads=4)
        elif case % 3 == 1:                                   # SRS path (tables when n >= 1024), with an offset
            srs = khip.Srs(cid, g)
            off = int(rng.integers(0, n))
            got, ginf = srs.msm(sc[: n - off], offset=off)
            want, winf = cref.msm(cid, g[off:], sc[: n - off], threads=4)
            srs.close()
        else:                                                 # batch of 3 over the same bases
            srs = khip.Srs(cid, g)
            batch = np.stack([sc, np.roll(sc, 1, axis=0), rand_fe_fast(rng, n)])
            gots, ginfs = srs.msm_batch(batch)
            srs.close()
            for j in range(3):
                w, wi = cref.msm(cid, g, batch[j], threads=4)
                assert bool(ginfs[j]) == wi and (wi or np.array_equal(gots[j], w)), (case, j)
            continue
        assert ginf == winf and (winf or np.array_equal(got, want)), case


@pytest.mark.parametrize("k", [5, 8])
def test_batches_of_five_to_eight_over_the_tables(khip, k):
    """The seven chunks of t (prover.rs:923-926) are one batch over the monomial basis: batches of up to eight take the lane-cooperative
    reduction kernels of the latency path; uniform scalars, and a skewed batch (most scalars equal) for the hot-bucket kernels."""
    rng = np.random.default_rng(4242 + k)
    n = 1 << 12
    g = cref.srs_generate(0, 3, n, threads=8)
    srs = khip.Srs(0, g)
    for skew in (False, True):
        cols = np.stack([rand_fe_fast(rng, n) for _ in range(k)])
        if skew:
            cols[:, : n - 37] = cols[0, 0]
        got, ginf = srs.msm_batch(cols)
        for j in range(k):
            w, wi = cref.msm(0, g, cols[j], threads=8)
            assert bool(ginf[j]) == wi and (wi or np.array_equal(got[j], w)), (k, skew, j)
    srs.close()


def test_ntt_randomized_differential(khip):
    rng = np.random.default_rng(777)
    for case in range(24):
        fid = case & 1
        logn = int(rng.integers(0, 15))
        batch = int(rng.integers(1, 6))
        x = rand_fe_fast(rng, batch << logn).reshape(batch, 1 << logn, 4)
        x[rng.random(x.shape[:2]) < 0.2] = 0
        inv = bool(rng.integers(0, 2))
        assert np.array_equal(khip.ntt(fid, x, logn, inv), cref.ntt(fid, x, logn, inv, threads=4)), (case, logn, batch, inv)
        if 1 <= logn <= 12:
            logb = int(rng.integers(1, 4))
            assert np.array_equal(khip.lde(fid, x, logn, logb), cref.lde(fid, x, logn, logb, threads=4)), (case, logn, logb)


@pytest.mark.parametrize("cid", [0, 1])
def test_msm_table_path_exceptional_cases(khip, cid):
    """Window-table path (n >= 1024) with bases that collide on purpose: the second half of the SRS repeats
    the first half (every bucket meets P + P: doubling branch of the mixed addition) or its negation
    (P + (-P): the accumulator passes through the identity and keeps going)."""
    F = P.CURVES[cid].scalar
    base_fid = 1 if cid == 0 else 0
    rng = np.random.default_rng(31 + cid)
    h = 1024
    g = cref.srs_generate(cid, 0, h, threads=8)
    sc_half = rand_fe(rng, h, F)
    sc = np.concatenate([sc_half, sc_half])
    dup = np.concatenate([g, g])
    neg = dup.copy()
    neg[h:, 4:] = cref.field_op(base_fid, "sub", np.zeros((h, 4), np.uint64), g[:, 4:])
    for pts, expect_inf in ((dup, False), (neg, True)):
        srs = khip.Srs(cid, pts)
        got, ginf = srs.msm(sc)
        want, winf = cref.msm(cid, pts, sc, threads=8)
        srs.close()
        assert ginf == winf == expect_inf and (winf or np.array_equal(got, want))
    # mixed: a third of the pairs cancel, a third double, the rest are unrelated
    mix = dup.copy()
    mix[h: h + h // 3] = neg[h: h + h // 3]
    mix[h + 2 * (h // 3):] = cref.srs_generate(cid, 5000, h - 2 * (h // 3), threads=8)
    srs = khip.Srs(cid, mix)
    got, ginf = srs.msm(sc)
    want, winf = cref.msm(cid, mix, sc, threads=8)
    srs.close()
    assert ginf == winf and np.array_equal(got, want)


@pytest.mark.parametrize("cid", [0, 1])
def test_degenerate_bases_through_the_tree_kernels(khip, cid):
    """One point repeated / two points / P, -P pairs as the whole basis, with 3-bit, 20-bit and full scalars, single and
    paired MSMs: every bucket, marginal and hot-bucket tree (incl. the lane-cooperative additions and their doubling
    fallback) meets equal points, cancellations and identities; results equal the C oracle's."""
    rng = np.random.default_rng(77 + cid)
    base = khip.srs_generate(cid, 0, 2)
    fid = 1 if cid == 0 else 0
    negp = base[0].copy()
    negp[4:] = cref.field_op(fid, "sub", np.zeros((1, 4), np.uint64), base[0, 4:].reshape(1, 4))[0]
    n = 1 << 12
    for other in (base[0], base[1], negp):
        g = np.tile(base[0], (n, 1)); g[1::2] = other
        srs = khip.Srs(cid, g)
        for bits in (253, 20, 3):
            sc = rng.integers(0, 1 << 63, size=(2 * n, 4), dtype=np.uint64)
            if bits <= 64:
                sc[:, 1:] = 0; sc[:, 0] &= np.uint64((1 << bits) - 1)
            else:
                sc[:, 3] &= np.uint64((1 << 61) - 1)
            d = khip.DevBuf(sc.nbytes).upload(sc)
            for k in (1, 2):
                got, ginf = srs.msm_batch_dev(d.ptr, n, k, mont=False)
                for j in range(k):
                    w, winf = cref.msm(cid, g, sc[j * n:(j + 1) * n], scalars_mont=False, threads=8)
                    assert bool(ginf[j]) == bool(winf) and (winf or np.array_equal(got[j], w)), (bits, k, j)
            d.free()
        srs.close()


def test_hip_path_against_derived_vectors(khip):
    """Fixed bytes committed under tests/golden/derived_vectors.json (definition-level big-int results)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "derived_vectors.json")) as f:
        d = json.load(f)
    for v in d["ntt"]:
        F = P.Fp if v["field"] == "Fp" else P.Fq
        fid = 0 if v["field"] == "Fp" else 1
        A = cref.ints_to_limbs([F.to_mont(int(x, 16)) for x in v["input"]])
        assert [F.from_mont(x) for x in cref.limbs_to_ints(khip.ntt(fid, A, v["log2_n"], False))] == [int(x, 16) for x in v["forward"]]
        assert [F.from_mont(x) for x in cref.limbs_to_ints(khip.ntt(fid, A, v["log2_n"], True))] == [int(x, 16) for x in v["inverse"]]
    for v in d["lde"]:
        F = P.Fp if v["field"] == "Fp" else P.Fq
        fid = 0 if v["field"] == "Fp" else 1
        c = cref.ints_to_limbs([F.to_mont(int(x, 16)) for x in v["coeffs"]])
        assert [F.from_mont(x) for x in cref.limbs_to_ints(khip.lde(fid, c, v["log2_n"], v["log2_blowup"]))] == [int(x, 16) for x in v["evals"]]
    for v in d["msm"]:
        c = P.CURVES[v["curve"]]
        g = khip.srs_generate(c.cid, 0, 12)
        sc = cref.ints_to_limbs([c.scalar.to_mont(int(x, 16)) for x in v["scalars"]])
        out, inf = khip.msm_points(c.cid, g, sc)
        assert not inf
        assert c.base.from_mont(P.from_limbs(out[:4])) == int(v["result"][0], 16)
        assert c.base.from_mont(P.from_limbs(out[4:])) == int(v["result"][1], 16)


@pytest.mark.parametrize("fid", [0, 1])
@pytest.mark.parametrize("logn", [20, 21, 22])
def test_ntt_large_sizes(khip, fid, logn):
    """2^20, 2^21, 2^22 directly against the oracle's DFT, forward and inverse, both fields (a transform with a wrong root of unity or a consistent
    output permutation passes a round trip and a linearity check: those are extras here, not the gate)."""
    rng = np.random.default_rng(4242 + 2 * logn + fid)
    n = 1 << logn
    x = rand_fe_fast(rng, n).reshape(1, n, 4)
    x[0, rng.integers(0, n, size=64)] = 0
    f = khip.ntt(fid, x, logn, False)
    assert np.array_equal(f, cref.ntt(fid, x, logn, False, threads=8)), ("forward", logn)
    i = khip.ntt(fid, x, logn, True)
    assert np.array_equal(i, cref.ntt(fid, x, logn, True, threads=8)), ("inverse", logn)
    assert np.array_equal(khip.ntt(fid, f, logn, True), x)


def test_ntt_large_properties(khip):
    """Size-independent properties at the largest shapes: linearity at 2^22, and the 2^18 -> 2^21 extension against the oracle and back."""
    rng = np.random.default_rng(4243)
    y = rand_fe_fast(rng, 1 << 22).reshape(1, 1 << 22, 4)
    z = rand_fe_fast(rng, 1 << 22).reshape(1, 1 << 22, 4)
    fy, fz = khip.ntt(0, y, 22, False), khip.ntt(0, z, 22, False)
    s = cref.field_op(0, "add", y.reshape(-1, 4), z.reshape(-1, 4)).reshape(1, 1 << 22, 4)
    assert np.array_equal(khip.ntt(0, s, 22, False).reshape(-1, 4), cref.field_op(0, "add", fy.reshape(-1, 4), fz.reshape(-1, 4)))
    c = rand_fe_fast(rng, 1 << 18).reshape(1, 1 << 18, 4)
    e = khip.lde(0, c, 18, 3)
    assert np.array_equal(e, cref.lde(0, c, 18, 3, threads=8))
    back = khip.ntt(0, e, 21, True)
    assert np.array_equal(back[:, : 1 << 18], c) and not back[:, 1 << 18:].any()


@pytest.fixture
def max_logr(khip, request):
    khip.set_ntt_max_logr(request.param)
    yield request.param
    khip.set_ntt_max_logr(0)


@pytest.mark.parametrize("max_logr", [8, 9, 10], indirect=True)
@pytest.mark.parametrize("fid", [0, 1])
def test_ntt_pass_shapes(khip, fid, max_logr):
    """The alternative pass decompositions (kh_ntt_set_max_logr / KH_NTT_MAX_LOGR: sub-transforms of up to 2^8, 2^9 = the default, 2^10 points) give the
    same bits: every size class of the splitter (one, two, three passes; even and ragged splits) for NTT, iNTT and the extension, against the oracle."""
    rng = np.random.default_rng(1000 * max_logr + fid)
    for logn in (3, 7, 8, 9, 10, 11, 13, 15, 16, 17, 18, 19, 20, 21):
        n = 1 << logn
        batch = 3 if logn <= 11 else (2 if logn <= 17 else 1)
        x = rand_fe_fast(rng, batch * n).reshape(batch, n, 4)
        for inverse in (False, True):
            assert np.array_equal(khip.ntt(fid, x, logn, inverse), cref.ntt(fid, x, logn, inverse, threads=8)), (max_logr, logn, inverse)
    for logn, logb in ((3, 3), (8, 2), (10, 3), (13, 1), (15, 3), (16, 3), (17, 3), (18, 3), (19, 1)):
        n = 1 << logn
        batch = 2 if logn <= 16 else 1
        c = rand_fe_fast(rng, batch * n).reshape(batch, n, 4)
        assert np.array_equal(khip.lde(fid, c, logn, logb), cref.lde(fid, c, logn, logb, threads=8)), (max_logr, logn, logb)


def test_ntt_max_logr_is_validated(khip):
    with pytest.raises(Exception):
        khip.set_ntt_max_logr(3)
    with pytest.raises(Exception):
        khip.set_ntt_max_logr(11)
    khip.set_ntt_max_logr(0)
