"""BASELINE.json configurations at FULL size inside `pytest -m gpu` ("next" item 1): every parity
claim DESIGN.md makes about the 2^20 / 2^22 MSMs is a collected test here, bit-exact against the C oracle
(oracle/pasta_ref.c: Jacobian Pippenger, pinned on the reference's vectors by tests/test_oracle_kats.py).

  * config 2: 2^20-point MSM over SRS::<Vesta>::create(2^20).g -- uniform scalars, the bench circuit's witness
    (n - 10 ones, 7 zeros, 3 random: kimchi/src/bench.rs:106), all-equal scalars, 20-bit scalars (the hot-bucket paths of
    csrc/msm.hip), Montgomery and canonical scalar input;
  * config 4: ONE 2^22-point MSM as 8 point-range shards of 2^19 (kh_srs_create_device_range per shard, kh_points_sum
    for the fold -- exactly what 8 ranks do, run one after the other on one GPU), against one oracle MSM;
  * the degenerate-basis stress (equal points, two points, P / -P pairs) at 2^12 and 2^16.
The bases come from the device generator, which tests/test_gpu_srs_trait.py pins on srs/vesta.srs; the oracle gets the
SAME points (downloaded), so a generator bug cannot hide a kernel bug or vice versa."""
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
    return k


@pytest.fixture(scope="module")
def vesta20(khip):
    srs = khip.Srs.create(khip.VESTA, 1 << 20)
    g = srs.get_g()
    yield srs, g
    srs.close()


def _rand_fe(rng, n):
    c = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    c[:, 3] &= np.uint64((1 << 61) - 1)                     # < 2^253 < p: a valid element in either representation
    return c


def _scalars(kind, n, rng):
    F = P.Fp
    one = cref.ints_to_limbs([F.R])
    if kind == "uniform":
        return _rand_fe(rng, n), True
    if kind == "uniform_canonical":
        return _rand_fe(rng, n), False
    if kind == "bench_witness":                             # kimchi/src/bench.rs:106 + the zero-knowledge rows
        return np.concatenate([np.repeat(one, n - 10, 0), np.zeros((7, 4), np.uint64), _rand_fe(rng, 3)]), True
    if kind == "all_equal":
        return np.repeat(_rand_fe(rng, 1), n, 0), True
    if kind == "bits20":
        s = np.zeros((n, 4), np.uint64)
        s[:, 0] = rng.integers(0, 1 << 20, n).astype(np.uint64)
        return s, False
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["uniform", "uniform_canonical", "bench_witness", "all_equal", "bits20"])
def test_config2_msm_2_20(khip, vesta20, kind):
    srs, g = vesta20
    n = 1 << 20
    sc, mont = _scalars(kind, n, np.random.default_rng(20 + len(kind)))
    got, ginf = srs.msm(sc, mont=mont)                                     # host buffers through kh_msm
    want, winf = cref.msm(0, g, sc, scalars_mont=mont, threads=THREADS)
    assert bool(ginf) == winf and (winf or np.array_equal(got, want)), kind
    buf = khip.DevBuf(sc.nbytes).upload(sc)                                # resident scalars, the benchmarked entry point
    got2, ginf2 = srs.msm_batch_dev(buf.ptr, n, 1, mont=mont)
    buf.free()
    assert bool(ginf2[0]) == winf and (winf or np.array_equal(got2[0], want)), kind


def test_config2_pallas_2_20(khip):
    """The other curve at the same size (config 5 keeps both resident): MSM<Fp coordinates>."""
    n = 1 << 20
    srs = khip.Srs.create(khip.PALLAS, n)
    g = srs.get_g()
    sc = _rand_fe(np.random.default_rng(77), n)
    got, ginf = srs.msm(sc)
    want, winf = cref.msm(1, g, sc, threads=THREADS)
    srs.close()
    assert bool(ginf) == winf and np.array_equal(got, want)


def test_config4_msm_2_22_as_8_shards(khip):
    """2^22 points, 8 ranks x 2^19: rank r owns g[r 2^19, (r+1) 2^19) and the matching scalars, emits one partial; the
    partials are folded with kh_points_sum (RCCL has no group-addition reduction: all-gather + local fold)."""
    R, per = 8, 1 << 19
    rng = np.random.default_rng(4)
    sc = _rand_fe(rng, R * per)
    parts = np.zeros((R, 8), np.uint64); pinf = np.zeros(R, np.uint8)
    gs = []
    for r in range(R):
        srs = khip.Srs.create(khip.VESTA, per, start=r * per)
        gs.append(srs.get_g())
        parts[r], inf = srs.msm(sc[r * per:(r + 1) * per])
        pinf[r] = inf
        srs.close()
    got, ginf = khip.points_sum(khip.VESTA, parts, pinf)
    g = np.concatenate(gs)
    assert np.array_equal(g[:4], cref.srs_generate(0, 0, 4, threads=1)) and np.array_equal(g[per:per + 2], cref.srs_generate(0, per, 2, threads=1))
    want, winf = cref.msm(0, g, sc, threads=THREADS)
    assert bool(ginf) == winf and np.array_equal(got, want)
    # and the same MSM unsharded on one GPU (config 4 on a single device)
    srs = khip.Srs.create(khip.VESTA, R * per)
    got1, inf1 = srs.msm(sc)
    srs.close()
    assert not inf1 and np.array_equal(got1, want)


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("logn", [12, 16])
def test_degenerate_bases_stress(khip, cid, logn):
    """MSMs over degenerate bases (one point repeated, two points, P / -P pairs) with full, 20-bit and 3-bit scalars force
    equal-point doublings, cancellations and identities through the accumulation (incl. the hand-over from the lazy
    29-bit kernel to the exact one), the bucket sums and the tree reductions; singles and k = 2 batches."""
    rng = np.random.default_rng(7 + cid + logn)
    n = 1 << logn
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
                for j in range(k):
                    w, winf = cref.msm(cid, g, scs[j * n:(j + 1) * n], scalars_mont=False, threads=8)
                    assert bool(ginf[j]) == bool(winf) and (winf or np.array_equal(got[j], w)), (variant, bits, k, j)
        srs.close()


# ---------------------------------------------------------------------------------------------- configs 3 / 5 and chunked proofs at size
def _accepted(khip, ix, proof, seed=9):
    from oracle import kimchi as K
    from oracle import views as V
    c, vix, pr = V.device_views(ix, proof)
    return K.verify(c, vix, pr, None, vix["h"], P.StdRng(bytes([seed] * 32)), final_msm=V.final_msm_c(c, ix.srs.get_g(), ix.size, threads=THREADS))


def test_config5_pallas_and_vesta_proofs_at_2_16_from_one_process(khip):
    """BASELINE config 5: the recursion pair -- a Vesta proof and a Pallas proof at 2^16 gates, both SRS (window tables, Lagrange
    bases) and all four kernel instantiations resident in ONE process, proved concurrently from two host threads; on devices 0 / 1
    when two GPUs are visible, else both on device 0.  Each proof is accepted by the oracle's restatement of the reference verifier
    (which runs the Fq-side multi-pass NTT kernels at 2^16 / 2^18 / 2^19 inside a Pallas proof for the first time in pytest)."""
    import threading
    from proof_systems_amd import prover
    ndev = max(1, khip.device_count())
    out, err = [None, None], []

    def run(k):
        try:
            khip.set_device(k % ndev)
            cid = (khip.VESTA, khip.PALLAS)[k]
            ix = prover.bench_circuit_index(cid, 16)
            F = prover.Fld(ix.fid)
            wit = np.tile(F.limbs(1), (15, (1 << 16) - 10, 1))
            proofs = [prover.create_proof(ix, wit, np.random.default_rng(40 + k + 2 * j)) for j in range(2)]
            out[k] = (ix, proofs)
        except Exception as e:                                  # noqa: BLE001 -- reported below
            err.append((k, repr(e)))
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    khip.set_device(0)
    for k in range(2):
        ix, proofs = out[k]
        assert ix.curve == (khip.VESTA, khip.PALLAS)[k] and ix.n == 1 << 16
        assert _accepted(khip, ix, proofs[1]), ("vesta", "pallas")[k]
        ix.free()


def test_chunked_proof_2_17_rows_over_the_2_16_srs(khip):
    """kimchi/src/tests/chunked.rs:91: a 2^17-row circuit proved over the 2^16 SRS -- num_chunks = 2, zk_rows = 5, every polynomial
    committed and evaluated in two chunks, 14 chunks of t, ft linearised with zeta^(2^16) (prover.rs:208-212, 989-1004, 1147-1188);
    accepted by the oracle's verifier (chunk-aware: verifier.rs:795-880)."""
    from proof_systems_amd import prover
    srs = khip.Srs.create(khip.VESTA, 1 << 16)
    ix = prover.bench_circuit_index(khip.VESTA, 17, srs=srs)
    assert (ix.num_chunks, ix.zk_rows, ix.size) == (2, 5, 1 << 16)
    F = prover.Fld(ix.fid)
    wit = np.tile(F.limbs(1), (15, (1 << 17) - 10, 1))
    proof = prover.create_proof(ix, wit, np.random.default_rng(17))
    assert len(proof["t_comm"][1]) == 14 and all(len(c[1]) == 2 for c in proof["w_comm"]) and len(proof["evals"]["z"][0]) == 2
    assert _accepted(khip, ix, proof)
    bad = dict(proof, evals=dict(proof["evals"], z=([proof["evals"]["z"][0][0], (proof["evals"]["z"][0][1] + 1) % F.p], proof["evals"]["z"][1])))
    assert not _accepted(khip, ix, bad)                          # the SECOND chunk's evaluation matters
    ix.free()


def test_config4_through_kh_msm_sharded(khip, vesta20):
    """BASELINE config 4 inside the LIBRARY (no torch, no Python threads): kh_msm_sharded over R handles created with
    kh_srs_create_device_range -- one per visible device round-robin (all on device 0 on a one-GPU box) -- equals ONE oracle MSM
    over the whole range; the device-resident form (kh_msm_sharded_dev: all jobs submitted before the first wait) gives the same
    point; fewer scalars than points and an empty tail shard are handled."""
    _, g = vesta20
    ndev = max(1, khip.device_count())
    for R, total in ((3, 3 << 16), (8, 1 << 20)):
        per = total // R
        shards = []
        for r in range(R):
            khip.set_device(r % ndev)
            shards.append(khip.Srs.create(khip.VESTA, per, start=r * per))
        khip.set_device(0)
        rng = np.random.default_rng(R)
        sc = _rand_fe(rng, R * per)
        want, winf = cref.msm(0, g[:R * per], sc, threads=THREADS)
        out, inf = khip.msm_sharded(shards, sc)
        assert inf == bool(winf) and np.array_equal(out, want), R
        bufs = []
        for r in range(R):
            khip.set_device(r % ndev)
            bufs.append(khip.DevBuf(per * 32).upload(sc[r * per:(r + 1) * per]))
        khip.set_device(0)
        out, inf = khip.msm_sharded_dev(shards, bufs, [per] * R)
        assert np.array_equal(out, want), R
        short = R * per - per - 5                               # the last shard gets nothing, the one before 5 scalars less
        want2, _ = cref.msm(0, g[:short], sc[:short], threads=THREADS)
        out, inf = khip.msm_sharded(shards, sc[:short])
        assert np.array_equal(out, want2)
        for b in bufs:
            b.free()
        for s in shards:
            s.close()
