"""The wide-window MSM path (csrc/msm.hip "wide windows": a second table set with 20-bit windows, 2^19 buckets, the lazy
two-plane bucket reduction), bit-exact against the C oracle.  By default the path serves single MSMs of >= 2^19 scalars
(tests/test_gpu_baseline_configs.py runs those at full size); here kh_msm_set_wide_min_n lowers the threshold so that
every branch runs in seconds: the five scalar distributions of config 2 at 2^14 on both curves, batches of two, an MSM
over a window of a longer basis, and the degenerate bases (one point repeated, two points, P / -P pairs) whose equal-point
doublings and cancellations force the hand-over lists of the lazy accumulation AND of the lazy reduction (add29 ->
k_wide_a1_exact) and the split / hot bucket branches of k_acc_wide_rest / k_bucket_sum_wide.  Reference behaviour: the MSM of
poly-commitment/src/ipa.rs:638-683 / commitment.rs:350-394 is a unique group element whatever the window width."""
import os

import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P

pytestmark = pytest.mark.gpu
THREADS = min(64, os.cpu_count() or 8)


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    k.set_wide_min_n(1 << 12)
    yield k
    k.set_wide_min_n(1 << 19)


def _rand_fe(rng, n):
    c = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    c[:, 3] &= np.uint64((1 << 61) - 1)
    return c


def _scalars(kind, n, rng):
    one = cref.ints_to_limbs([P.Fp.R])
    if kind == "uniform":
        return _rand_fe(rng, n), True
    if kind == "uniform_canonical":
        return _rand_fe(rng, n), False
    if kind == "bench_witness":
        return np.concatenate([np.repeat(one, n - 10, 0), np.zeros((7, 4), np.uint64), _rand_fe(rng, 3)]), True
    if kind == "all_equal":
        return np.repeat(_rand_fe(rng, 1), n, 0), True
    if kind == "bits20":
        s = np.zeros((n, 4), np.uint64)
        s[:, 0] = rng.integers(0, 1 << 20, n).astype(np.uint64)
        return s, False
    if kind == "top_window":                                  # only the 13th window (bits 240..254) is populated
        s = np.zeros((n, 4), np.uint64)
        s[:, 3] = rng.integers(0, 1 << 61, n).astype(np.uint64) & np.uint64(((1 << 61) - 1) ^ ((1 << 48) - 1))
        return s, False
    raise KeyError(kind)


def _wide_ran(khip):
    return any(name == "reduce_a1" for name, _ in khip.last_timings())


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("kind", ["uniform", "uniform_canonical", "bench_witness", "all_equal", "bits20", "top_window"])
def test_wide_msm_distributions(khip, cid, kind):
    n = 1 << 14
    srs = khip.Srs.create(cid, n)
    g = srs.get_g()
    sc, mont = _scalars(kind, n, np.random.default_rng(3 + cid + len(kind)))
    got, ginf = srs.msm(sc, mont=mont)
    want, winf = cref.msm(cid, g, sc, scalars_mont=mont, threads=THREADS)
    assert bool(ginf) == winf and (winf or np.array_equal(got, want)), kind
    buf = khip.DevBuf(sc.nbytes).upload(sc)
    got2, ginf2 = srs.msm_batch_dev(buf.ptr, n, 1, mont=mont)
    assert _wide_ran(khip), "the wide path did not serve this MSM"
    assert bool(ginf2[0]) == winf and (winf or np.array_equal(got2[0], want)), kind
    # a window of the basis: offset 1000, 2^13 scalars
    m = 1 << 13
    got3, ginf3 = srs.msm(sc[:m], mont=mont, offset=1000)
    want3, winf3 = cref.msm(cid, g[1000:1000 + m], sc[:m], scalars_mont=mont, threads=THREADS)
    buf.free(); srs.close()
    assert bool(ginf3) == winf3 and (winf3 or np.array_equal(got3, want3)), kind


def test_wide_msm_batch_of_two_and_below_threshold(khip):
    n = 1 << 13
    srs = khip.Srs.create(khip.VESTA, n)
    g = srs.get_g()
    sc = _rand_fe(np.random.default_rng(5), 2 * n)
    d = khip.DevBuf(sc.nbytes).upload(sc)
    got, ginf = srs.msm_batch_dev(d.ptr, n, 2)
    assert _wide_ran(khip)
    for j in range(2):
        w, winf = cref.msm(0, g, sc[j * n:(j + 1) * n], threads=THREADS)
        assert not ginf[j] and not winf and np.array_equal(got[j], w)
    # fewer scalars than the threshold: the narrow tables of the same handle
    m = 1 << 11
    got, ginf = srs.msm_batch_dev(d.ptr, m, 1)
    assert not _wide_ran(khip)
    w, winf = cref.msm(0, g[:m], sc[:m], threads=THREADS)
    d.free(); srs.close()
    assert not ginf[0] and np.array_equal(got[0], w)


def test_wide_msm_ragged_sizes_and_digit_boundaries(khip):
    """Sizes that are no power of two (the sort's blocks and runs end anywhere), and scalars built from the boundary digits of the signed 20-bit
    decomposition: windows equal to 2^19 (the largest positive digit), 2^19 + 1 (the first that turns negative and carries), 2^20 - 1, 0, plus p - 1 and 1."""
    p_s = P.Fp.p
    rng = np.random.default_rng(99)
    for n in (4097, 5000, 12345):
        srs = khip.Srs.create(khip.VESTA, n)
        g = srs.get_g()
        vals = []
        for i in range(n):
            if i < 4:
                vals.append([0, 1, p_s - 1, p_s - 2][i])
                continue
            v = 0
            for w in range(13):
                d = [1 << 19, (1 << 19) + 1, (1 << 20) - 1, 0, int(rng.integers(0, 1 << 20))][int(rng.integers(0, 5))]
                v |= d << (20 * w)
            vals.append(v % p_s)
        sc = cref.ints_to_limbs(vals)
        got, ginf = srs.msm(sc, mont=False)
        assert _wide_ran(khip)
        want, winf = cref.msm(0, g, sc, scalars_mont=False, threads=THREADS)
        srs.close()
        assert bool(ginf) == winf and (winf or np.array_equal(got, want)), n


@pytest.mark.parametrize("cid", [0, 1])
def test_wide_degenerate_bases(khip, cid):
    rng = np.random.default_rng(70 + cid)
    n = 1 << 12
    base = khip.srs_generate(cid, 0, 8)
    fid = 1 if cid == 0 else 0

    def rs(k, bits):
        a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64)
        if bits <= 64:
            a[:, 1:] = 0; a[:, 0] &= np.uint64((1 << bits) - 1)
        else:
            a[:, 3] &= np.uint64((1 << 61) - 1)
        return a

    for variant in ("all_same", "two_points", "pairs_opposite"):
        g = np.tile(base[0], (n, 1))
        if variant == "two_points":
            g[1::2] = base[1]
        if variant == "pairs_opposite":
            neg = base[0].copy()
            neg[4:] = cref.field_op(fid, "sub", np.zeros((1, 4), np.uint64), base[0, 4:].reshape(1, 4))[0]
            g[1::2] = neg
        srs = khip.Srs(cid, g)
        for bits in (253, 20, 3):
            sc = rs(n, bits)
            for k in (1, 2):
                scs = np.concatenate([sc, sc[::-1]]) if k == 2 else sc
                d = khip.DevBuf(scs.nbytes).upload(scs)
                got, ginf = srs.msm_batch_dev(d.ptr, n, k, mont=False)
                d.free()
                assert _wide_ran(khip)
                for j in range(k):
                    w, winf = cref.msm(cid, g, scs[j * n:(j + 1) * n], scalars_mont=False, threads=8)
                    assert bool(ginf[j]) == bool(winf) and (winf or np.array_equal(got[j], w)), (variant, bits, k, j)
        srs.close()


def test_wide_tables_are_optional(khip):
    """the 20-bit-window table set is a throughput optimisation, not part of a handle's contract -- kh_srs_set_wide_tables gives it back
    (the narrow tables then serve the same MSM, same group element) and builds it again on request; kh_srs_has_wide_tables reports the state."""
    n = 1 << 13
    rng = np.random.default_rng(99)
    g = cref.srs_generate(0, 0, n, threads=THREADS)
    sc = _rand_fe(rng, n)
    want, winf = cref.msm(0, g, sc, threads=THREADS)
    srs = khip.Srs(khip.VESTA, g)
    assert srs.has_wide_tables()
    got, ginf = srs.msm(sc)
    assert _wide_ran(khip) and ginf == winf and np.array_equal(got, want)
    srs.set_wide_tables(False)
    assert not srs.has_wide_tables()
    got, ginf = srs.msm(sc)
    assert not _wide_ran(khip) and ginf == winf and np.array_equal(got, want)
    srs.set_wide_tables(True)
    assert srs.has_wide_tables()
    got, ginf = srs.msm(sc)
    assert _wide_ran(khip) and ginf == winf and np.array_equal(got, want)
    srs.close()
    # a basis too small for any window tables cannot get the wide set either
    small = khip.Srs(khip.VESTA, g[:512])
    assert not small.has_wide_tables()
    with pytest.raises(Exception):
        small.set_wide_tables(True)
    small.close()


@pytest.mark.parametrize("stage,passes", [(0, 2), (2560, 8), (4096, 3), (3000, 1), (28672, 2)])
def test_wide_sort_staging_settings(khip, stage, passes):
    """k_part2_sort's staged scatter (round 6: a partition's entries are collected in LDS and written out as coalesced runs, in passes over consecutive bucket
    ranges) under settings that force, at 2^15..2^16 scalars (1.7-3.3 K entries per partition): several passes, the fall-back to the direct scatter when a partition would
    need more passes than allowed, buckets longer than the 1024-entry margin that straddle the end of the staging area (all_equal / bench_witness: one bucket
    holds a window's every entry) and the direct scatter itself.  Same group element under every setting."""
    khip.set_sort_staging(stage, passes)
    try:
        for n, kinds in ((1 << 15, ["uniform", "all_equal", "bench_witness", "top_window"]), (1 << 16, ["uniform", "bits20"]), (40000, ["uniform"])):
            srs = khip.Srs.create(khip.VESTA, n)
            g = srs.get_g()
            for kind in kinds:
                sc, mont = _scalars(kind, n, np.random.default_rng(stage + passes + len(kind)))
                want, winf = cref.msm(0, g, sc, scalars_mont=mont, threads=THREADS)
                for rep in range(3):                              # (the third call replays the captured launch sequence)
                    got, ginf = srs.msm(sc, mont=mont)
                    assert _wide_ran(khip) or rep > 0
                    assert bool(ginf) == winf and (winf or np.array_equal(got, want)), (kind, n, rep)
            srs.close()
    finally:
        khip.set_sort_staging()


def test_sort_staging_rejects_too_many_passes(khip):
    with pytest.raises(Exception):
        khip.set_sort_staging(4096, 9)
    khip.set_sort_staging()
