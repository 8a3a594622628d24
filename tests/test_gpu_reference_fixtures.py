"""The device path against bytes the REFERENCE produced (tests/golden/ref_fixtures/, see tests/test_reference_fixtures.py):
for the circuit of kimchi's `test_generic_gate` (polynomials/generic.rs:380-470) the device-built prover index has the reference's
verifier-index commitments and digest; the device iNTT (+ commitment over the monomial basis), the 8x extension (+ strided
commitment over the Lagrange basis) and the chunk evaluator reproduce the reference's commitments and the evaluations its proof
states at its own zeta / zeta omega; and a proof made by the device prover for this circuit is accepted by the oracle verifier
under the index commitments taken from the reference's bytes.  Rows a7 / a8 / a12: pinned on reference-generated data."""
import os

import numpy as np
import pytest

from oracle import cref
from oracle import fixtures as FX
from oracle import kimchi as K
from oracle import pasta as P
from oracle import views as V

from test_gpu_prover import _aff, _verify
from test_reference_fixtures import C, F, HERE, generic_test_circuit

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def _limbs(vals):
    return cref.ints_to_limbs([F.to_mont(v) for v in vals])


def _ints(limbs):
    return [F.from_mont(v) for v in cref.limbs_to_ints(np.asarray(limbs).reshape(-1, 4))]


def test_device_index_transforms_and_prover_against_the_reference_bytes(khip):
    from proof_systems_amd import prover
    fx = FX.load(os.path.join(HERE, "test_generic_gate.bin"), C)
    v = fx["vindex"]
    rows, wit = generic_test_circuit()
    co = np.stack([_limbs(r) for r in rows])                              # (20, 15, 4)
    ix = prover.ProverIndex(khip.VESTA, 5, co)
    one = lambda t: V.chunks(C, t)
    # ---- the index commitments and the digest
    assert [one(t) for t in ix.sigma_comm] == v["sigma_comm"]
    assert [one(t) for t in ix.coefficients_comm] == v["coefficients_comm"]
    assert one(ix.generic_comm) == v["generic_comm"] and one(ix.zero_selector_comm) == v["psm_comm"]
    h = C.srs_h()
    vix_ref = dict(v); vix_ref["h"] = h
    assert C.base.from_mont(P.from_limbs(ix.digest)) == K.verifier_index_digest(C, vix_ref)
    assert ix.shifts == v["shifts"] and ix.omega == v["omega"]
    # ---- the transforms on the reference's columns: sigma_3 and coefficient column 8 (rebuilt by the oracle, whose commitments
    #      were checked against the same bytes on the CPU)
    oix = K.build_index(F, 5, rows)
    srs = ix.srs
    fid = khip.FP
    vixv, proofv = FX.oracle_views(fx, h)
    ch = K.fiat_shamir(C, vixv, proofv, K.verifier_index_digest(C, vixv))
    zeta = ch["zeta"]; zetaw = zeta * v["omega"] % F.p
    ev = proofv["evals"]
    cases = [(oix["sigma"][3], v["sigma_comm"][3], ev["s"][3]), (oix["coefficients"][8], v["coefficients_comm"][8], ev["coefficients"][8]),
             (oix["coefficients"][0], v["coefficients_comm"][0], ev["coefficients"][0])]
    for col, want_comm, want_eval in cases:
        e = _limbs(col)
        coeffs = khip.ntt(fid, e[None], 5, inverse=True)[0]                # Evaluations::interpolate
        com, inf = srs.commit_non_hiding(coeffs, 1)
        assert [_aff(C, com[0], inf[0])] == want_comm
        d8 = khip.lde(fid, coeffs[None], 5, 3)[0]                          # evaluate_over_domain_by_ref(d8)
        com, inf = srs.commit_evaluations_non_hiding(5, d8)
        assert [_aff(C, com[0], inf[0])] == want_comm
        assert np.array_equal(khip.ntt(fid, d8[None].copy(), 8, inverse=True)[0][:32], coeffs) and not khip.ntt(fid, d8[None].copy(), 8, inverse=True)[0][32:].any()
        buf = khip.DevBuf(32 * 32).upload(coeffs)
        got = khip.evaluate_chunks_dev(fid, buf, 32, 32, 1, _limbs([zeta, zetaw]))
        assert tuple(_ints(got)) == tuple(want_eval)                       # the evaluations the reference's proof states
        buf.free()
    # ---- the device prover on the reference's circuit and witness; verifier index commitments from the reference's bytes
    w = np.stack([_limbs(c) for c in wit])                                 # (15, 20, 4)
    proof = prover.create_proof(ix, w, np.random.default_rng(4))
    ok, (c, vix, pr) = _verify(khip, ix, proof)
    assert ok
    for k in ("sigma_comm", "coefficients_comm", "generic_comm", "psm_comm", "complete_add_comm", "mul_comm", "emul_comm", "endomul_scalar_comm"):
        vix[k] = v[k]                                                       # (equal anyway: asserted above)
    g_l = ix.srs.get_g()
    final_msm = V.final_msm_c(C, g_l, ix.size, threads=8)
    assert K.verify(C, vix, pr, None, h, P.StdRng(bytes([7] * 32)), final_msm=final_msm)


def test_device_prover_on_the_reference_public_input_circuit(khip):
    """kimchi's `test_generic_gate_pub` (5 public inputs = 3, then create_circuit's 20 generic rows): the device-built index has the
    reference's commitments, and the device prover's proof is accepted with the public commitment the VERIFIER computes from the
    inputs (verifier.rs:834-858) -- a different public input is rejected."""
    from proof_systems_amd import prover
    from test_reference_fixtures import lagrange_commitments
    fx = FX.load(os.path.join(HERE, "test_generic_gate_pub.bin"), C)
    v = fx["vindex"]
    rows, wit = generic_test_circuit()
    pub_rows = [[1] + [0] * 14 for _ in range(5)]                          # GenericGateSpec::Pub: coefficient 1 on the left wire
    co = np.stack([_limbs(r) for r in pub_rows + rows])
    ix = prover.ProverIndex(khip.VESTA, 5, co, public=5)
    one = lambda t: V.chunks(C, t)
    assert [one(t) for t in ix.sigma_comm] == v["sigma_comm"] and [one(t) for t in ix.coefficients_comm] == v["coefficients_comm"]
    assert one(ix.generic_comm) == v["generic_comm"] and v["public"] == 5
    w = [[3] * 5 + col for col in wit]
    for c in range(1, 15):
        w[c][:5] = [0] * 5
    proof = prover.create_proof(ix, np.stack([_limbs(c) for c in w]), np.random.default_rng(6))
    assert proof["evals"]["public"] != ([0], [0])
    h = C.srs_h()
    g_l = ix.srs.get_g()
    lag = lagrange_commitments(g_l, 5, 5)
    from test_gpu_prover import _oracle_views

    final_msm = V.final_msm_c(C, g_l, ix.size, threads=8)
    for public, want in (([3] * 5, True), ([3, 3, 4, 3, 3], False)):
        c, vix, pr = _oracle_views(khip, ix, proof)
        vix["public_comm"] = K.public_commitment(C, h, lag, public)
        assert K.verify(C, vix, pr, None, h, P.StdRng(bytes([3] * 32)), final_msm=final_msm) == want
        # the evaluations of the public polynomial the proof carries are the ones the verifier would compute from the inputs
        if want:
            ch = K.fiat_shamir(C, vix, pr, K.verifier_index_digest(C, vix))
            assert K.public_evaluations(F, ix.n, ix.omega, public, ch["zeta"]) == tuple(x[0] for x in proof["evals"]["public"])


FIXTURES_WITH_GATE_TERMS = ["test_poseidon", "ec_test", "varbase_mul_test", "endomul_test", "endomul_scalar_test", "test_prove_and_verify_xor", "and_prove_and_verify_vesta",
                            "verify_range_check_valid_proof1", "rot_prove_and_verify_vesta", "test_ffadd_finalization", "test_max_foreign_multiplicands", "test_carry_plookups"]
FIXTURES_WITH_LOOKUPS = ["lookup_gate_proving_works", "lookup_gate_proving_works_multiple_tables", "test_prove_and_verify_xor", "verify_range_check_valid_proof1",
                         "test_max_foreign_multiplicands", "test_runtime_table"]


@pytest.mark.parametrize("name", sorted(set(FIXTURES_WITH_GATE_TERMS + FIXTURES_WITH_LOOKUPS)))
def test_device_token_programs_on_the_reference_proof_evaluations(khip, name):
    """The token programs of proof_systems_amd/polish.py, run by kh_expr_evaluations_dev on two-row columns made of the evaluations a
    REFERENCE proof states (row 0: at zeta, row 1: at zeta omega), give the constant term with which the oracle verifier accepts that
    proof (tests/test_reference_fixtures.py): the gate library, the optional gates (Xor16, RangeCheck0/1, Rot64, ForeignFieldAdd/Mul)
    and the lookup constraints incl. runtime tables, device side, on reference-generated data."""
    from proof_systems_amd import polish as OP
    from oracle import lookup as L
    fx = FX.load(os.path.join(HERE, name + ".bin"), C)
    h = C.srs_h()
    vix, proof = FX.oracle_views(fx, h)
    ch = K.fiat_shamir(C, vix, proof, K.verifier_index_digest(C, vix))
    ev = proof["evals"]
    alpha, zeta = ch["alpha"], ch["zeta"]
    fid = khip.FP
    col = lambda e: khip.DevBuf(64).upload(_limbs([e[0], e[1]]))
    wcols = [col(e) for e in ev["w"]]
    out = khip.DevBuf(32)
    if name in FIXTURES_WITH_GATE_TERMS:
        ccols = [col(e) for e in ev["coefficients"]]
        total = 0
        endo = P.endos(P.PALLAS)[0]
        sels = [(g, ev[key]) for g, key in K.GATE_SELECTORS] + [(g, e) for g, e in zip(K.OPTIONAL_GATES, ev["optional_gate_selectors"]) if e is not None]
        for gname, sel in sels:
            toks, consts = OP.gate_program(gname, F.p, alpha, selector_col=30, mds=OP.POSEIDON_MDS[0], endo=endo)
            khip.expr_evaluations_dev(fid, toks, wcols + ccols + [col(sel)], [2] * 31, _limbs(consts), 1, out, stride=1, next_shift=1)
            total = (total + _ints(out.download((1, 4)))[0]) % F.p
        assert total == K.gate_library_constant_term(C, ev, alpha) and total != 0
    if name in FIXTURES_WITH_LOOKUPS:
        li = vix["lookup_index"]
        n, omega, zk = vix["n"], vix["omega"], vix["zk_rows"]
        jc = ch["joint_combiner"]
        one = lambda x: khip.DevBuf(32).upload(_limbs([x]))
        c = 15
        cols = {"sorted": list(range(c, c + li["max_per_row"] + 1))}
        c += li["max_per_row"] + 1
        cols["aggreg"], cols["table"] = c, c + 1
        cols["selector"] = {q: c + 2 + k for k, q in enumerate(li["patterns"])}
        c += 2 + len(li["patterns"])
        cols["vanish"], cols["l0"], cols["lfinal"] = c, c + 1, c + 2
        bufs = wcols + [col(e) for e in ev["lookup_sorted"]] + [col(ev["lookup_aggregation"]), col(ev["lookup_table"])]
        bufs += [col(ev["lookup_selectors"][q]) for q in li["patterns"]]
        bufs += [one(L.vanishes_on_last_n_rows(F.p, omega, n, zk + 1, zeta)), one(L.unnormalized_lagrange_basis(F.p, omega, n, 0, zeta)),
                 one(L.unnormalized_lagrange_basis(F.p, omega, n, -(zk + 1), zeta))]
        lens = [2] * (len(bufs) - 3) + [1] * 3
        if li["uses_runtime_tables"]:
            cols["runtime"], cols["runtime_selector"] = c + 3, c + 4
            bufs += [col(ev["runtime_lookup_table"]), col(ev["runtime_lookup_table_selector"])]
            lens += [2, 2]
        toks, consts = OP.lookup_program(F.p, li["patterns"], cols, jc, pow(jc, li["max_joint_size"], F.p), ch["beta"], ch["gamma"], alpha, alpha0=K.ALPHA_LOOKUP0)
        khip.expr_evaluations_dev(fid, toks, bufs, lens, _limbs(consts), 1, out, stride=1, next_shift=1)
        assert _ints(out.download((1, 4)))[0] == K.lookup_constant_term(F, vix, ev, ch, zeta)
