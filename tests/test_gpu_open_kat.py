"""The reference's opening-proof known-answer test (poly-commitment/tests/commitment.rs:388-440) with the device
doing the work: the 11 commitments through kh_commit_non_hiding, combine_polys and b_init through
kh_combine_polys_dev / kh_b_init_dev, the 7 folding rounds through kh_ipa_begin_dev + kh_ipa_round_*; the oracle
only supplies what stays on the host in the reference integration as well (RNG, sponge, transcript, serialisation)."""
import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P
from oracle import poseidon as S

pytestmark = pytest.mark.gpu


def _aff(c, xy, inf):
    if inf:
        return None
    return (c.base.from_mont(P.from_limbs(xy[:4])), c.base.from_mont(P.from_limbs(xy[4:])))


def _limbs(F, vals):
    return cref.ints_to_limbs([F.to_mont(v) for v in vals])


def test_opening_proof_kat_on_device(golden):
    import proof_systems_amd.khip as khip
    khip.init(0)
    k = golden["opening_proof_kat"]
    cid = 0; c = P.CURVES[cid]; F = c.scalar
    n = k["srs_depth"]
    srs = khip.Srs.create(cid, n)                       # SRS::create on the device
    g = [_aff(c, row, 0) for row in srs.get_g(0, n)]
    h = _aff(c, khip.srs_h(cid), 0)

    def commit(coeffs):
        if not coeffs:
            return [None]
        xy, inf = srs.commit_non_hiding(_limbs(F, coeffs), 1)
        return [_aff(c, xy[j], inf[j]) for j in range(len(inf))]

    def device_vectors(plnms, polyscale, elm, evalscale, n_):
        bufs, lens, chunks = [], [], []
        for coeffs, blinders in plnms:
            d = khip.DevBuf(max(len(coeffs), 1) * 32)
            if coeffs:
                d.upload(_limbs(F, coeffs))
            bufs.append(d); lens.append(len(coeffs)); chunks.append(len(blinders))
        a_dev = khip.DevBuf(n_ * 32); b_dev = khip.DevBuf(n_ * 32)
        plen = khip.combine_polys_dev(khip.FP, bufs, lens, chunks, _limbs(F, [polyscale])[0], n_, a_dev)
        khip.b_init_dev(khip.FP, _limbs(F, elm), _limbs(F, [evalscale])[0], n_, b_dev)
        p_all = [F.from_mont(v) for v in cref.limbs_to_ints(a_dev.download((n_, 4)))]
        assert not any(p_all[plen:])
        b = [F.from_mont(v) for v in cref.limbs_to_ints(b_dev.download((n_, 4)))]
        for d in bufs:
            d.free()
        return p_all[:plen], b, (a_dev, b_dev)

    class DeviceRounds:
        def __init__(self, a, b, u_base, handle):
            u_l = cref.ints_to_limbs([c.base.to_mont(u_base[0]), c.base.to_mont(u_base[1])]).reshape(8)
            a_dev, b_dev = handle
            self.op = khip.IpaOpening(srs, a_dev, b_dev, u_l, a_len=n, b_len=n)
            a_dev.free(); b_dev.free()
        def round_lr(self, rand_l, rand_r):
            xy, inf = self.op.round_lr(_limbs(F, [rand_l])[0], _limbs(F, [rand_r])[0])
            return _aff(c, xy[0], inf[0]), _aff(c, xy[1], inf[1])
        def round_fold(self, u_pre):
            u, _ = self.op.round_fold(u_pre)
            return F.from_mont(P.from_limbs(u))
        def finish(self):
            a0, b0, sg, sginf = self.op.finish()
            self.op.free()
            return F.from_mont(P.from_limbs(a0)), F.from_mont(P.from_limbs(b0)), _aff(c, sg, sginf)

    proof, _ = P.first_random_opening_proof(c, g, h, P.StdRng(bytes(k["seed"])), S.DefaultFqSponge(c),
                                            commit=commit, rounds_backend=DeviceRounds, vectors_backend=device_vectors)
    buf = P.msgpack_opening_proof(c, proof)
    want = bytes(k["bytes"])
    assert buf == want[:len(buf)] and not any(want[len(buf):])
    srs.close()


def test_batch_verifier_msm_on_device(golden):
    """SRS::verify (ipa.rs:301-502) on two proofs (the KAT proof and one from another seed): the transcript side runs
    in the oracle, the one big MSM through kh_ipa_verify_msm (challenge polynomials expanded on the device and
    multiplied into the resident tables).  Valid batch -> zero; a tampered z1 or a swapped L -> non-zero."""
    import proof_systems_amd.khip as khip
    khip.init(0)
    cid = 0; c = P.CURVES[cid]; F = c.scalar
    n = 128
    srs = khip.Srs.create(cid, n)
    g = [_aff(c, row, 0) for row in srs.get_g(0, n)]
    h = _aff(c, khip.srs_h(cid), 0)
    items = []
    for seed in (bytes(32), bytes([7] * 32)):
        proof, _ = P.first_random_opening_proof(c, g, h, P.StdRng(seed), S.DefaultFqSponge(c))
        items.append(proof["verifier_input"])

    def device_verify(batch):
        fresh = [dict(it, sponge=it["sponge"].clone()) for it in batch]
        g_terms, pts, sc = P.ipa_verify_terms(c, n, h, fresh, P.StdRng(bytes([3] * 32)))
        chals = sum((ch for _, ch in g_terms), [])
        xy = np.stack([cref.ints_to_limbs([c.base.to_mont(p[0]), c.base.to_mont(p[1])]).reshape(8) if p else np.zeros(8, np.uint64) for p in pts])
        inf = np.array([0 if p else 1 for p in pts], dtype=np.uint8)
        return khip.ipa_verify_msm(srs, _limbs(F, chals), _limbs(F, [w for w, _ in g_terms]), xy, _limbs(F, sc), inf)

    assert device_verify(items)
    assert device_verify(items[:1])
    assert P.ipa_verify(c, g, h, [dict(it, sponge=it["sponge"].clone()) for it in items], P.StdRng(bytes([3] * 32)))
    bad = dict(items[1], opening=dict(items[1]["opening"], z1=(items[1]["opening"]["z1"] + 1) % F.p))
    assert not device_verify([items[0], bad])
    lr = list(items[0]["opening"]["lr"]); lr[2] = (lr[2][1], lr[2][0])
    assert not device_verify([dict(items[0], opening=dict(items[0]["opening"], lr=lr))])
    srs.close()
