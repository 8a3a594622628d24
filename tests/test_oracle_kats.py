"""Pins both CPU oracles (oracle/pasta.py big-int, oracle/pasta_ref.c) against the
reference's own golden vectors (tests/golden/reference_kats.json, extracted from
/root/reference by tests/golden/make_golden.py) and against each other."""
import hashlib
import random

import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P


def _pt_limbs(curve, pts):
    F = curve.base
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    inf = np.zeros(len(pts), dtype=np.uint8)
    for i, pt in enumerate(pts):
        if pt is None:
            inf[i] = 1
            continue
        out[i, :4] = P.to_limbs(F.to_mont(pt[0]))
        out[i, 4:] = P.to_limbs(F.to_mont(pt[1]))
    return out, inf


def _pt_from_limbs(curve, xy, inf=False):
    if inf:
        return None
    F = curve.base
    return (F.from_mont(P.from_limbs(xy[:4])), F.from_mont(P.from_limbs(xy[4:])))


def _sc_limbs(F, vals, mont=True):
    return cref.ints_to_limbs([F.to_mont(v) if mont else v for v in vals])


# ---------------------------------------------------------------- constants
@pytest.mark.parametrize("name,F", [("Fp", P.Fp), ("Fq", P.Fq)])
def test_field_constants(golden, name, F):
    g = golden["fields"][name]
    assert F.p == int(g["modulus_dec"]) == int(g["MODULUS"], 16)
    assert F.R == int(g["R"], 16)
    assert F.R2 == int(g["R2"], 16)
    assert F.inv64 == int(g["INV"])
    assert F.t == int(g["T"], 16)
    assert F.to_mont(F.two_adic_root) == int(g["TWO_ADIC_ROOT_OF_UNITY"], 16)
    assert F.to_mont(P.GENERATOR) == int(g["GENERATOR"], 16)
    # C oracle derives the same constants on its own
    import ctypes as C
    p = np.zeros(4, np.uint64); one = np.zeros(4, np.uint64); r2 = np.zeros(4, np.uint64); root = np.zeros(4, np.uint64)
    inv = C.c_uint64(0)
    fid = 0 if name == "Fp" else 1
    cref.lib().ko_field_consts(fid, cref._p64(p), cref._p64(one), cref._p64(r2), C.byref(inv), cref._p64(root))
    assert P.from_limbs(p) == F.p and P.from_limbs(one) == F.R and P.from_limbs(r2) == F.R2
    assert inv.value == F.inv64 and P.from_limbs(root) == int(g["TWO_ADIC_ROOT_OF_UNITY"], 16)


def test_generators(golden):
    for name, c in (("vesta", P.VESTA), ("pallas", P.PALLAS)):
        gx, gy = (int(v) for v in golden["generators"][name])
        assert c.gen == (gx, gy) and c.is_on_curve(c.gen)
        assert c.mul(c.gen, c.scalar.p) is None            # cofactor 1, order = scalar modulus
        xy, _ = _pt_limbs(c, [c.gen])
        assert cref.lib().ko_is_on_curve(c.cid, cref._p64(xy)) == 1


# ---------------------------------------------------------------- field arithmetic C vs big-int
@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_c_field_ops_vs_bigint(fid, F):
    rnd = random.Random(1234 + fid)
    edge = [0, 1, 2, F.p - 1, F.p - 2, (F.p - 1) // 2, (1 << 254), F.R, F.R2 % F.p]
    a = edge + [rnd.randrange(F.p) for _ in range(500)]
    b = list(reversed(edge)) + [rnd.randrange(F.p) for _ in range(500)]
    A = cref.ints_to_limbs(a); B = cref.ints_to_limbs(b)
    Rinv = pow(1 << 256, -1, F.p)
    assert cref.limbs_to_ints(cref.field_op(fid, "mul", A, B)) == [x * y * Rinv % F.p for x, y in zip(a, b)]
    assert cref.limbs_to_ints(cref.field_op(fid, "add", A, B)) == [(x + y) % F.p for x, y in zip(a, b)]
    assert cref.limbs_to_ints(cref.field_op(fid, "sub", A, B)) == [(x - y) % F.p for x, y in zip(a, b)]
    assert cref.limbs_to_ints(cref.field_op(fid, "to_mont", A)) == [F.to_mont(x) for x in a]
    assert cref.limbs_to_ints(cref.field_op(fid, "from_mont", A)) == [F.from_mont(x) for x in a]
    nz = [x for x in a if x]
    inv = cref.limbs_to_ints(cref.field_op(fid, "inv", cref.ints_to_limbs([F.to_mont(x) for x in nz])))
    assert [F.from_mont(v) for v in inv] == [F.inv(x) for x in nz]


# ---------------------------------------------------------------- MSM KAT (kimchi/src/proof.rs:1160-1204)
def _b_poly_coefficients(chals, p):
    return P.b_poly_coefficients(P.Fp if p == P.Fp.p else P.Fq, chals)


def test_msm_kat(golden):
    kat = golden["msm_kat"]
    c = P.VESTA
    coeffs = _b_poly_coefficients(kat["chals"], c.scalar.p)
    assert coeffs == [1, 7, 5, 35, 3, 21, 15, 105, 2, 14, 10, 70, 6, 42, 30, 210]
    basis = [c.mul(c.gen, i) for i in range(1, 17)]
    want = (int(kat["expected_x"]), int(kat["expected_y"]))
    assert c.msm(basis, coeffs) == want
    assert c.msm_naive(basis, coeffs) == want
    xy, _ = _pt_limbs(c, basis)
    for mont in (True, False):
        sc = _sc_limbs(c.scalar, coeffs, mont)
        for naive in (False, True):
            out, inf = cref.msm(c.cid, xy, sc, scalars_mont=mont, naive=naive)
            assert not inf and _pt_from_limbs(c, out) == want


# ---------------------------------------------------------------- SRS (srs/*.srs; ipa.rs:751-778)
@pytest.mark.parametrize("name", ["vesta", "pallas"])
def test_srs_generator_matches_reference_file(golden, name):
    c = P.CURVES[name]
    g = golden["srs"][name]
    assert g["n"] == 65536
    # python oracle: sampled points + h
    for idx in ("0", "1", "2", "63", "1000", "65535"):
        assert c.compress(c.srs_g(int(idx))).hex() == g["samples"][idx]
    assert c.compress(c.srs_h()).hex() == g["h"]
    # C oracle: every sample, and the digest of the first 2^12 compressed points
    pts = cref.srs_generate(c.cid, 0, 4096, threads=8)
    comp = cref.compress(c.cid, pts)
    for idx, hx in g["samples"].items():
        i = int(idx)
        if i < 4096:
            assert bytes(comp[i]).hex() == hx
        else:
            one = cref.srs_generate(c.cid, i, 1)
            assert bytes(cref.compress(c.cid, one)[0]).hex() == hx
    assert hashlib.blake2b(comp.tobytes(), digest_size=32).hexdigest() == g["prefix_digest_blake2b256"]["12"]
    assert bytes(cref.compress(c.cid, cref.srs_h(c.cid).reshape(1, 8))[0]).hex() == g["h"]
    # decompress round trip
    assert c.decompress(bytes.fromhex(g["samples"]["5"])) == c.srs_g(5)


@pytest.mark.parametrize("name", ["vesta", "pallas"])
def test_srs_generator_full_digest(golden, name):
    """All 65,536 points of srs/{vesta,pallas}.srs reproduced by the C generator."""
    c = P.CURVES[name]
    pts = cref.srs_generate(c.cid, 0, 65536, threads=8)
    comp = cref.compress(c.cid, pts)
    assert hashlib.blake2b(comp.tobytes(), digest_size=32).hexdigest() == golden["srs"][name]["prefix_digest_blake2b256"]["16"]


# ---------------------------------------------------------------- trusted-setup KAT (tests/commitment.rs:289-345)
def test_trusted_setup_kat(golden):
    rng = P.StdRng(bytes(32))
    for name, F in (("vesta", P.Fp), ("pallas", P.Fq)):
        c = P.CURVES[name]
        x = P.field_rand(F, rng)
        buf = bytes(golden["srs_trusted_setup_kat"][name])
        assert buf[0] == 0x92 and buf[1] == 0x98
        pts = [buf[2 + 35 * i + 2: 2 + 35 * i + 35] for i in range(8)]
        h = buf[2 + 35 * 8 + 2: 2 + 35 * 8 + 35]
        g = c.gen
        for i in range(8):
            assert c.compress(c.mul(g, pow(x, i, F.p))) == pts[i], (name, i)
        assert c.compress(c.srs_h()) == h


# ---------------------------------------------------------------- commit KAT (tests/commitment.rs:348-386)
def test_commit_kat(golden):
    kat = golden["commit_kat"]
    c = P.VESTA
    rng = P.StdRng(bytes(kat["seed"]))
    g = c.srs_create(kat["srs_depth"])
    coeffs = [P.field_rand(P.Fp, rng) for _ in range(kat["com_length"] + 1)]
    com = P.commit_non_hiding(c, g, coeffs, kat["num_chunks"])
    assert com[3] is None and com[4] is None and com[5] is None
    blinders = [P.field_rand(P.Fp, rng) for _ in range(kat["num_chunks"])]
    out = P.mask_custom(c, c.srs_h(), com, blinders)
    got = P.msgpack_polycomm(c, out)
    want = bytes(kat["bytes"])          # the reference's expected buffer is zero-padded past the encoding
    assert want[:len(got)] == got and not any(want[len(got):])
    # the same three chunk MSMs through the C oracle
    gxy = cref.srs_generate(c.cid, 0, 128)
    for j in range(3):
        sc = _sc_limbs(P.Fp, coeffs[128 * j: 128 * (j + 1)])
        o, inf = cref.msm(c.cid, gxy, sc)
        assert _pt_from_limbs(c, o, inf) == com[j]


# ---------------------------------------------------------------- C Pippenger vs naive / big-int, edge cases
@pytest.mark.parametrize("cid", [0, 1])
def test_c_msm_edge_cases(cid):
    c = P.CURVES[cid]
    F = c.scalar
    rnd = random.Random(77 + cid)
    n = 300
    gxy = cref.srs_generate(cid, 0, n)
    pts = [_pt_from_limbs(c, gxy[i]) for i in range(n)]
    cases = {
        "uniform": [rnd.randrange(F.p) for _ in range(n)],
        "bench_circuit": [1] * (n - 10) + [0] * 7 + [rnd.randrange(F.p) for _ in range(3)],
        "zeros": [0] * n,
        "minus_one": [F.p - 1] * n,
        "small": [rnd.randrange(1 << 16) for _ in range(n)],
        "top_window": [(F.p - 1) - rnd.randrange(1 << 20) for _ in range(n)],
    }
    for name, sc in cases.items():
        want = c.msm(pts, sc)
        out, inf = cref.msm(cid, gxy, _sc_limbs(F, sc), threads=3)
        assert _pt_from_limbs(c, out, inf) == want, name
        out2, inf2 = cref.msm(cid, gxy[:40], _sc_limbs(F, sc[:40]), naive=True)
        assert _pt_from_limbs(c, out2, inf2) == c.msm(pts[:40], sc[:40]), name
    # repeated points, P and -P in one bucket, infinity inputs
    rep = [pts[0], pts[0], c.neg(pts[0]), pts[1], None, pts[1]]
    sc = [5, 5, 5, 9, 11, F.p - 9]
    xy, inf = _pt_limbs(c, rep)
    out, oinf = cref.msm(cid, xy, _sc_limbs(F, sc), inf=inf)
    assert _pt_from_limbs(c, out, oinf) == c.mul(pts[0], 5)
    out, oinf = cref.msm(cid, xy[:3], _sc_limbs(F, [1, 0, 1]), inf=inf[:3])
    assert oinf


# ---------------------------------------------------------------- NTT: definition, round trip, domain identities
@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_ntt_definition_and_roundtrip(fid, F):
    rnd = random.Random(5 + fid)
    for k in (0, 1, 2, 3, 6):
        n = 1 << k
        a = [rnd.randrange(F.p) for _ in range(n)]
        assert P.ntt(F, a, k) == P.dft_naive(F, a, k)
        assert P.ntt(F, P.ntt(F, a, k), k, inverse=True) == a
        A = cref.ints_to_limbs([F.to_mont(v) for v in a])
        fw = cref.ntt(fid, A, k, False)
        assert [F.from_mont(v) for v in cref.limbs_to_ints(fw)] == P.ntt(F, a, k)
        bw = cref.ntt(fid, fw, k, True)
        assert [F.from_mont(v) for v in cref.limbs_to_ints(bw)] == a
    # omega_{2^16} quoted in SURVEY.md 8(a9)
    w16 = F.root_of_unity(16)
    assert pow(w16, 1 << 16, F.p) == 1 and pow(w16, 1 << 15, F.p) == F.p - 1
    assert hex(w16).startswith("0x23222d06" if fid == 0 else "0x385e22fc")


@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_domain_nesting_and_lde(fid, F):
    """kimchi/tests/test_domain.rs:10-71: d1 subset d8 with stride 8; the d8
    evaluation of a degree<n polynomial sub-sampled with stride 8 is its d1 evaluation."""
    rnd = random.Random(9 + fid)
    k = 5
    n = 1 << k
    assert pow(F.root_of_unity(k + 3), 8, F.p) == F.root_of_unity(k)       # domains.rs:64-66
    coeffs = [rnd.randrange(F.p) for _ in range(n)]
    e8 = P.lde(F, coeffs, k, 3)
    assert e8[::8] == P.ntt(F, coeffs, k)
    assert P.ntt(F, e8[::8], k, inverse=True) == coeffs
    assert P.ntt(F, e8, k + 3, inverse=True) == coeffs + [0] * (7 * n)
    C8 = cref.lde(fid, cref.ints_to_limbs([F.to_mont(v) for v in coeffs]), k, 3)
    assert [F.from_mont(v) for v in cref.limbs_to_ints(C8)] == e8
    # batch + threads
    batch = np.stack([cref.ints_to_limbs([F.to_mont(rnd.randrange(F.p)) for _ in range(n)]) for _ in range(5)])
    got = cref.ntt(fid, batch, k, False, threads=4)
    for b in range(5):
        assert np.array_equal(got[b], cref.ntt(fid, batch[b], k, False)[0])


# ---------------------------------------------------------------- Lagrange basis (tests/ipa_commitment.rs:26-119)
def test_lagrange_basis_identity():
    c = P.VESTA
    F = c.scalar
    k = 3
    n = 1 << k
    g = c.srs_create(n)
    basis = P.lagrange_basis(c, g, k)
    for i in range(n):
        e = [0] * n
        e[i] = 1
        coeffs = P.ntt(F, e, k, inverse=True)
        assert basis[i] == P.commit_non_hiding(c, g, coeffs, 1)
    gxy = cref.srs_generate(c.cid, 0, n)
    bxy, binf = cref.lagrange_basis(c.cid, gxy, k)
    assert [_pt_from_limbs(c, bxy[i], binf[i]) for i in range(n)] == [b[0] for b in basis]
    # chunked: domain 2n over an SRS of size n -> 2 chunks per basis element (ipa_commitment.rs:54-86)
    basis2 = P.lagrange_basis(c, g, k + 1)
    for i in (0, 5, 2 * n - 1):
        e = [0] * (2 * n)
        e[i] = 1
        coeffs = P.ntt(F, e, k + 1, inverse=True)
        assert basis2[i] == P.commit_non_hiding(c, g, coeffs, 2)
    for ch in range(2):
        bxy, binf = cref.lagrange_basis(c.cid, gxy, k + 1, chunk=ch)
        assert [_pt_from_limbs(c, bxy[i], binf[i]) for i in range(2 * n)] == [b[ch] for b in basis2]
    # commit_evaluations == commit(interpolate(evals))
    rnd = random.Random(3)
    ev = [rnd.randrange(F.p) for _ in range(n)]
    assert P.commit_evaluations_non_hiding(c, basis, ev, k) == P.commit_non_hiding(c, g, P.ntt(F, ev, k, inverse=True), 1)


# ---------------------------------------------------------------- derived fixed vectors (tests/golden/make_derived.py)
def _derived():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "derived_vectors.json")) as f:
        return json.load(f)


def test_c_oracle_against_derived_vectors():
    d = _derived()
    for v in d["ntt"]:
        F = P.Fp if v["field"] == "Fp" else P.Fq
        fid = 0 if v["field"] == "Fp" else 1
        a = [int(x, 16) for x in v["input"]]
        A = cref.ints_to_limbs([F.to_mont(x) for x in a])
        assert [F.from_mont(x) for x in cref.limbs_to_ints(cref.ntt(fid, A, v["log2_n"], False))] == [int(x, 16) for x in v["forward"]]
        assert [F.from_mont(x) for x in cref.limbs_to_ints(cref.ntt(fid, A, v["log2_n"], True))] == [int(x, 16) for x in v["inverse"]]
        assert P.ntt(F, a, v["log2_n"]) == [int(x, 16) for x in v["forward"]]
    for v in d["lde"]:
        F = P.Fp if v["field"] == "Fp" else P.Fq
        fid = 0 if v["field"] == "Fp" else 1
        c = cref.ints_to_limbs([F.to_mont(int(x, 16)) for x in v["coeffs"]])
        got = cref.lde(fid, c, v["log2_n"], v["log2_blowup"])
        assert [F.from_mont(x) for x in cref.limbs_to_ints(got)] == [int(x, 16) for x in v["evals"]]
    for v in d["msm"]:
        c = P.CURVES[v["curve"]]
        g = cref.srs_generate(c.cid, 0, 12)
        sc = _sc_limbs(c.scalar, [int(x, 16) for x in v["scalars"]])
        out, inf = cref.msm(c.cid, g, sc)
        assert not inf and _pt_from_limbs(c, out) == (int(v["result"][0], 16), int(v["result"][1], 16))


@pytest.mark.parametrize("c", [P.VESTA, P.PALLAS], ids=["vesta", "pallas"])
def test_endos_and_scalar_challenge(c):
    """endos::<G>() (ipa.rs:214-231), ScalarChallenge::to_field (sponge.rs:190-226) and combine_one_endo
    (combine.rs:292-340).  The reference holds no literal vector for these constants; they are pinned by the
    properties the reference itself asserts (primitive cube roots of unity; phi(G) = [endo_r] G on the generator
    pinned by test_generators) and by the ladder == scalar equivalence its prover/verifier pair relies on."""
    eq, er = P.endos(c)
    assert eq != 1 and pow(eq, 3, c.base.p) == 1
    assert er != 1 and pow(er, 3, c.scalar.p) == 1
    g = c.gen
    assert c.mul(g, er) == (g[0] * eq % c.base.p, g[1])
    rnd = random.Random(5)
    pts = [c.mul(g, rnd.randrange(1, c.scalar.p)) for _ in range(4)]
    for chal in (rnd.getrandbits(128), 0, (1 << 128) - 1):
        k = P.challenge_to_field(c.scalar, chal, er)
        want = [c.add(a, c.mul(b, k)) for a, b in zip(pts[:2], pts[2:])]
        assert P.combine_one_endo(c, pts[:2], pts[2:], chal) == want


@pytest.mark.parametrize("cid", [0, 1])
def test_host_endos_match_oracle(cid):
    """kh_endos is host-only (no device): the C side's endo pair equals the oracle's."""
    from proof_systems_amd import khip
    c = P.CURVES[cid]
    q, r = khip.endos(cid)
    assert (c.base.from_mont(P.from_limbs(q)), c.scalar.from_mont(P.from_limbs(r))) == P.endos(c)


def test_poseidon_sponge_kats():
    """The oracle's Kimchi Poseidon / Fq-sponge against the reference's own pins: poseidon/tests/test_vectors/kimchi.json
    and the empty-challenge regressions (poseidon/tests/poseidon_tests.rs:74-108)."""
    import json, os
    from oracle import poseidon as S
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon_kimchi_params.json")))["kats"]
    assert len(k["kimchi_fp_hash"]) >= 5
    for v in k["kimchi_fp_hash"]:
        sp = S.ArithmeticSponge(P.Fp)
        sp.absorb([int.from_bytes(bytes.fromhex(x), "little") for x in v["input"]])
        assert sp.squeeze() == int.from_bytes(bytes.fromhex(v["output"]), "little")
    assert S.DefaultFqSponge(P.VESTA).challenge() == int.from_bytes(bytes.fromhex(k["challenge_empty_vesta"]), "little")
    assert S.DefaultFqSponge(P.PALLAS).challenge() == int.from_bytes(bytes.fromhex(k["challenge_empty_pallas"]), "little")


def test_opening_proof_kat(golden):
    """SRS::open end to end (ipa.rs:811-1063) through the oracle reproduces the reference's opening-proof bytes
    (tests/commitment.rs:388-440): pins endos, ScalarChallenge::to_field, combine_one_endo, the round structure,
    the RNG draw order and the sponge."""
    from oracle import poseidon as S
    k = golden["opening_proof_kat"]
    c = P.VESTA
    g = c.srs_create(k["srs_depth"]); h = c.srs_h()
    proof, _ = P.first_random_opening_proof(c, g, h, P.StdRng(bytes(k["seed"])), S.DefaultFqSponge(c))
    buf = P.msgpack_opening_proof(c, proof)
    want = bytes(k["bytes"])
    assert buf == want[:len(buf)] and not any(want[len(buf):])


def test_verifier_accepts_reference_proof(golden):
    """The oracle's SRS::verify (ipa.rs:301-502) accepts the byte-pinned opening proof of the reference's KAT and
    rejects it with one scalar changed -- the reference's own test_randomised property (tests/commitment.rs:233-260)."""
    from oracle import poseidon as S
    c = P.VESTA
    g = c.srs_create(128); h = c.srs_h()
    rng = P.StdRng(bytes(golden["opening_proof_kat"]["seed"]))
    proof, _ = P.first_random_opening_proof(c, g, h, rng, S.DefaultFqSponge(c))
    vi = proof["verifier_input"]
    assert P.ipa_verify(c, g, h, [dict(vi, sponge=vi["sponge"].clone())], rng)
    bad = dict(vi, sponge=vi["sponge"].clone(), opening=dict(proof, z2=(proof["z2"] + 1) % c.scalar.p))
    assert not P.ipa_verify(c, g, h, [bad], rng)


# ---- the identities kh_ipa_open's split of sg relies on (csrc/api.hip: ipa_sg_collect, csrc/ipa.hip: k_sg_split), against the literal definitions

def test_sg_splits_on_the_last_challenge():
    """sg = <b_poly_coefficients(chals), G> (ipa.rs:452-470) = A + [u_last] B with A / B the sums over the even / odd points weighted by the
    coefficient vector of the EARLIER challenges: what lets the device compute A and B underneath the last round."""
    curve = P.VESTA
    F = curve.scalar
    rng = random.Random(11)
    rounds = 4
    g = [curve.mul(curve.gen, rng.randrange(1, F.p)) for _ in range(1 << rounds)]
    chals = [rng.randrange(1, F.p) for _ in range(rounds)]          # chals[j] = the challenge of round j + 1
    sg = curve.msm(g, P.b_poly_coefficients(F, chals))
    # the folded basis of the literal loop (g <- g_lo + u g_hi, ipa.rs:1006) ends at the same point
    gg = list(g)
    for u in chals:
        m = len(gg) // 2
        gg = [curve.add(gg[i], curve.mul(gg[i + m], u)) for i in range(m)]
    assert gg[0] == sg
    # device convention (k_ipa_fold): coef' [2i] = coef[i], coef'[2i+1] = coef[i] u, so the last challenge sits on the lowest index bit
    coef = [1]
    for u in chals[:-1]:
        coef = [c * f % F.p for c in coef for f in (1, u)]
    A = curve.msm(g[0::2], coef)
    B = curve.msm(g[1::2], coef)
    assert curve.add(A, curve.mul(B, chals[-1])) == sg


def test_endo_challenge_decomposition():
    """u = challenge_to_field(c) = a endo_r + b with a, b < 2^67 (sponge.rs:190-226), hence [u] B = [a] phi(B) + [b] B with
    phi(x, y) = (endo_q x, y): the 67-step joint ladder the host uses instead of a 255-bit one."""
    curve = P.VESTA
    F = curve.scalar
    eq, er = P.endos(curve)
    rng = random.Random(12)
    Bp = curve.mul(curve.gen, rng.randrange(1, F.p))
    for _ in range(4):
        c = rng.getrandbits(128)
        a = b = 2
        for i in reversed(range(64)):
            a, b = 2 * a, 2 * b
            s = 1 if (c >> (2 * i)) & 1 else -1
            if (c >> (2 * i + 1)) & 1:
                a += s
            else:
                b += s
        assert 0 < a < 1 << 67 and 0 < b < 1 << 67
        u = P.challenge_to_field(F, c, er)
        assert (a * er + b) % F.p == u
        phi = (Bp[0] * eq % curve.base.p, Bp[1])
        assert curve.add(curve.mul(phi, a), curve.mul(Bp, b)) == curve.mul(Bp, u)
