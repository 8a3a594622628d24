"""The reference's OWN stored proofs (kimchi/src/tests/fixtures/*.bin: proof + verifier index serialised by the reference's
prover, verified by its prover-less test mode, tests/generic.rs:56-103) as golden vectors.  Copies of ALL FORTY live in
tests/golden/ref_fixtures/ (binary test data, read with oracle/fixtures.py); nothing here touches /root/reference.  Every one of
them is accepted by the oracle's verifier: generic gates, the gate library, lookups, Xor16, range checks, Rot64, foreign-field
addition and multiplication, runtime tables, recursion, and the two Pallas proofs.

What they pin, none of it "by definition":
  * the oracle's VERIFIER (oracle/kimchi.py) accepts every one of these reference-generated proofs and rejects tampered ones --
    generic gates with and without public inputs, and one circuit per gate type of the library (Poseidon, CompleteAdd,
    VarBaseMul, EndoMul, EndoMulScalar), whose constraint rows (oracle/gates.py) enter ft_eval0 as the linearization's
    constant term: the row machines the device token programs are tested against are the reference's constraints;
  * the lookup argument: two proofs of circuits of 500 Lookup gates (one table / five tables with ids) are accepted with
    oracle/lookup.py's constraint values as the lookup part of the constant term, the combined table commitment of combine_table
    and the transcript / evaluation order of verifier.rs:179-246, 1034-1175;
  * the index side: domain generator, Shifts::new, sigma, the coefficient layout of create_generic_gadget, and -- through the
    Lagrange-basis commitments of the index columns -- the inverse-DFT conventions of the oracle's NTT (rows a7 / a8 of
    SURVEY 8): the rebuilt commitments equal the reference's bytes.
No GPU."""
import os

import numpy as np
import pytest

from oracle import cref
from oracle import fixtures as FX
from oracle import kimchi as K
from oracle import pasta as P

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
C = P.VESTA
F = C.scalar
SRS_LEN = 65536                                   # every fixture was made over the reference's 2^16 test SRS

GATE_FIXTURES = {"test_poseidon": "poseidon_selector", "test_poseidon_in_circuit_extra_zero_block": "poseidon_selector", "ec_test": "complete_add_selector", "varbase_mul_test": "mul_selector",
                 "endomul_test": "emul_selector", "endomul_scalar_test": "endomul_scalar_selector"}
GENERIC_FIXTURES = ["test_generic_gate", "test_generic_gate_pub", "test_generic_gate_pub_empty", "test_generic_gate_pub_all_zeros",
                    "test_prove_and_verify_five_not_gnrc"]                   # (the Not gadget built from generic gates, tests/not.rs)
XOR_FIXTURES = ["and_prove_and_verify_vesta", "test_prove_and_verify_xor", "test_xor_finalization", "test_prove_and_verify_not_xor"]   # Xor16 + its lookups
# the optional gates (range_check/, rot.rs, foreign_field_add/, foreign_field_mul/): their row machines (oracle/gates.py) are the constant
# term with which these proofs verify; runtime tables (lookup/runtime_tables.rs); previous challenges (tests/recursion.rs); Pallas
OPTIONAL_GATE_FIXTURES = {"verify_range_check_valid_proof1": (0, 1), "verify_compact_multi_range_check_proof": (0, 1), "rot_prove_and_verify_vesta": (0, 5), "test_rot_finalization": (0, 5),
                          "test_ffadd_finalization": (0, 1, 2), "prove_and_verify_1": (2,), "prove_and_verify_50": (2,),
                          "test_zero_mul": (0, 1, 3), "test_one_mul": (0, 1, 3), "test_max_native_square": (0, 1, 3), "test_max_foreign_square": (0, 1, 3),
                          "test_max_native_multiplicands": (0, 1, 3), "test_max_foreign_multiplicands": (0, 1, 3),
                          "test_carry_plookups": (3,), "test_invalid_carry1_bit": (3,), "test_invalid_wraparound_carry1_hi": (3,)}    # indices into OPTIONAL_GATES
RUNTIME_FIXTURES = ["test_runtime_table", "test_runtime_table_only_one_table_with_id_zero_with_non_zero_entries_fixed_values",
                    "test_runtime_table_only_one_table_with_id_zero_with_non_zero_entries_random_values"]
PALLAS_FIXTURES = ["and_prove_and_verify_pallas", "rot_prove_and_verify_pallas"]
LOOKUP_FIXTURES = ["lookup_gate_proving_works", "lookup_gate_proving_works_multiple_tables",     # tests/lookup.rs:38-170: 500 Lookup gates
                   "test_dummy_value_is_added_in_an_arbitraly_created_table_when_no_table_with_id_0"]   # the dummy entry's table is synthesised (lookup/index.rs)


def generic_test_circuit():
    """polynomials/generic.rs:380-429 (create_circuit(0, 0)) and :436-470 (fill_in_witness): coefficient rows and witness."""
    p = F.p
    rows = [[1, 3, p - 1, 0, 0, 0, 0, p - 1, 2, 0] + [0] * 5 for _ in range(10)] + [[1, 0, 0, 0, p - 3, 1, 0, 0, 0, p - 5] + [0] * 5 for _ in range(10)]
    wit = [[0] * 20 for _ in range(15)]
    for r in range(10):
        wit[0][r], wit[1][r], wit[2][r] = 11, 23, 11 + 23 * 3
        wit[3][r], wit[4][r], wit[5][r] = 11, 23, 11 * 23 * 2
    for r in range(10, 20):
        wit[0][r], wit[3][r] = 3, 5
    return rows, wit


@pytest.fixture(scope="module")
def srs():
    return cref.srs_generate(0, 0, SRS_LEN, threads=8), C.srs_h()


def lagrange_commitments(g_l, log2_n, count):
    """commitments to L_0 .. L_{count-1} of the domain over the first n SRS points (one chunk): n^-1 sum_j w^(-ij) g_j"""
    n = 1 << log2_n
    winv = F.inv(F.root_of_unity(log2_n)); ninv = F.inv(n)
    out = []
    for i in range(count):
        sc = cref.ints_to_limbs([F.to_mont(ninv * pow(winv, i * j, F.p) % F.p) for j in range(n)])
        xy, inf = cref.msm(0, g_l[:n], sc, threads=8)
        assert not inf
        v = cref.limbs_to_ints(xy.reshape(2, 4))
        out.append((C.base.from_mont(v[0]), C.base.from_mont(v[1])))
    return out


def final_msm_for(g_l, n_srs, curve=C):
    """the verifier's one MSM (ipa.rs:452-502), in the C oracle"""
    Fs = curve.scalar
    cid = 0 if curve is P.VESTA else 1

    def final_msm(g_terms, pts, sc):
        gs = [0] * n_srs
        for w, chal in g_terms:
            for j, s in enumerate(P.b_poly_coefficients(Fs, chal)):
                gs[j] = (gs[j] + w * s) % Fs.p
        live = [(p, s) for p, s in zip(pts, sc) if p is not None]
        xy = np.concatenate([g_l[:n_srs], np.stack([cref.ints_to_limbs([curve.base.to_mont(p[0]), curve.base.to_mont(p[1])]).reshape(8) for p, _ in live])])
        scal = cref.ints_to_limbs([Fs.to_mont(s) for s in gs + [s for _, s in live]])
        _, inf = cref.msm(cid, xy, scal, threads=8)
        return inf
    return final_msm


def verify(fx, srs, mutate=None, curve=C):
    g_l, h = srs
    vix, proof = FX.oracle_views(fx, h)
    n_srs = vix["max_poly_size"]
    if fx["public"]:
        assert curve is C
        vix["public_comm"] = K.public_commitment(C, h, lagrange_commitments(g_l, vix["log2_n"], len(fx["public"])), fx["public"])
    if mutate:
        mutate(vix, proof)
    return K.verify(curve, vix, proof, None, h, P.StdRng(bytes([5] * 32)), final_msm=final_msm_for(g_l, n_srs, curve))


def test_every_stored_fixture_of_the_reference_is_covered():
    names = sorted(f[:-4] for f in os.listdir(HERE) if f.endswith(".bin"))
    covered = GENERIC_FIXTURES + list(GATE_FIXTURES) + LOOKUP_FIXTURES + XOR_FIXTURES + list(OPTIONAL_GATE_FIXTURES) + RUNTIME_FIXTURES + PALLAS_FIXTURES + ["test_recursion"]
    assert names == sorted(covered) and len(names) == 40


@pytest.mark.parametrize("name", list(OPTIONAL_GATE_FIXTURES) + RUNTIME_FIXTURES + ["test_recursion"])
def test_oracle_verifier_accepts_optional_gate_runtime_table_and_recursive_proofs(name, srs):
    fx = FX.load(os.path.join(HERE, name + ".bin"), C)
    v, ev = fx["vindex"], fx["proof"]["evals"]
    if name in OPTIONAL_GATE_FIXTURES:
        on = [k for k, e in enumerate(ev["optional_gate_selectors"]) if e is not None]
        assert tuple(on) == OPTIONAL_GATE_FIXTURES[name] == tuple(k for k, c in enumerate(v["optional_comms"]) if c is not None)
    if name in RUNTIME_FIXTURES:
        assert v["lookup_index"]["uses_runtime_tables"] and fx["proof"]["lookup"]["runtime"] is not None and ev["runtime_lookup_table"] is not None
    if name == "test_recursion":
        assert len(fx["proof"]["prev_challenges"]) == 1 and len(fx["proof"]["prev_challenges"][0][0]) == 16
    assert verify(fx, srs)

    def bump_w(vix, proof):                          # every gate type of these circuits reads witness column 1 (and the permutation does)
        w = list(proof["evals"]["w"]); w[1] = ((w[1][0] + 1) % F.p, w[1][1]); proof["evals"]["w"] = w
    assert not verify(fx, srs, bump_w)


@pytest.mark.parametrize("name", PALLAS_FIXTURES)
def test_oracle_verifier_accepts_the_pallas_proofs(name):
    """the same circuits proved over Pallas (scalar field Fq, sponge parameters of the other curve): and.rs / rot.rs `test_prove_and_verify`"""
    CP = P.PALLAS
    fx = FX.load(os.path.join(HERE, name + ".bin"), CP)
    srs_p = (cref.srs_generate(1, 0, SRS_LEN, threads=8), CP.srs_h())
    assert fx["endo"] is None or fx["endo"] == P.endos(P.VESTA)[0]
    assert verify(fx, srs_p, curve=CP)


@pytest.mark.parametrize("name", GENERIC_FIXTURES + list(GATE_FIXTURES) + LOOKUP_FIXTURES + XOR_FIXTURES)
def test_oracle_verifier_accepts_the_reference_proof(name, srs):
    fx = FX.load(os.path.join(HERE, name + ".bin"), C)
    v = fx["vindex"]
    assert v["max_poly_size"] == SRS_LEN and v["zk_rows"] == 3
    assert (v["lookup_index"] is not None) == (name in LOOKUP_FIXTURES + XOR_FIXTURES) == (fx["proof"]["lookup"] is not None)
    if name in XOR_FIXTURES:
        assert v["lookup_index"]["patterns"] == ["Xor"] and fx["proof"]["evals"]["optional_gate_selectors"][4] is not None
    if name in LOOKUP_FIXTURES:
        li = v["lookup_index"]
        assert li["patterns"] == ["Lookup"] and (li["max_per_row"], li["max_joint_size"], li["joint_lookup_used"]) == (3, 2, True)
        assert (li["table_ids"] is not None) == (not name.endswith("proving_works")) and len(fx["proof"]["lookup"]["sorted"]) == 4
    assert v["omega"] == F.root_of_unity(v["log2_n"]) and v["shifts"] == K.sample_shifts(F, v["log2_n"])        # domains.rs, Shifts::new
    if name in GATE_FIXTURES:                      # the gate type under test is live in this proof: its selector does not evaluate to 0
        assert fx["proof"]["evals"][GATE_FIXTURES[name]][0][0] != 0
    if fx["endo"] is not None:
        assert fx["endo"] == P.endos(P.PALLAS)[0]
    assert verify(fx, srs)


def test_oracle_verifier_rejects_tampering(srs):
    fx = FX.load(os.path.join(HERE, "test_poseidon.bin"), C)

    def bump_eval(vix, proof):
        z = proof["evals"]["z"]; proof["evals"]["z"] = (z[0], (z[1] + 1) % F.p)

    def bump_ft(vix, proof):
        proof["ft_eval1"] = (proof["ft_eval1"] + 1) % F.p

    def swap_sigma(vix, proof):
        vix["sigma_comm"] = list(vix["sigma_comm"]); vix["sigma_comm"][2], vix["sigma_comm"][3] = vix["sigma_comm"][3], vix["sigma_comm"][2]

    def bump_witness_eval(vix, proof):            # breaks the Poseidon constant term only
        w = list(proof["evals"]["w"]); w[4] = ((w[4][0] + 1) % F.p, w[4][1]); proof["evals"]["w"] = w
    for m in (bump_eval, bump_ft, swap_sigma, bump_witness_eval):
        assert not verify(fx, srs, m), m.__name__
    fxl = FX.load(os.path.join(HERE, "lookup_gate_proving_works_multiple_tables.bin"), C)

    def bump_sorted(vix, proof):                  # breaks the lookup constant term only
        srt = list(proof["evals"]["lookup_sorted"]); srt[2] = ((srt[2][0] + 1) % F.p, srt[2][1]); proof["evals"]["lookup_sorted"] = srt

    def bump_aggreg(vix, proof):
        a = proof["evals"]["lookup_aggregation"]; proof["evals"]["lookup_aggregation"] = (a[0], (a[1] + 1) % F.p)

    def drop_table_ids(vix, proof):               # the combined table commitment loses its table-id term
        vix["lookup_index"] = dict(vix["lookup_index"]); vix["lookup_index"]["table_ids"] = None
    for m in (bump_sorted, bump_aggreg, drop_table_ids):
        assert not verify(fxl, srs, m), m.__name__
    fxp = FX.load(os.path.join(HERE, "test_generic_gate_pub.bin"), C)
    fxp["public"][2] = 4                            # a different public input
    assert not verify(fxp, srs)


def test_rebuilt_index_commitments_equal_the_reference_bytes():
    """The verifier index of create_circuit(0, 0), rebuilt from the circuit with the oracle -- constraint-system columns
    (oracle/kimchi.py::build_index), the Lagrange basis by definition AND the oracle's iNTT + monomial-basis commitment --
    equals the commitments the reference serialised."""
    fx = FX.load(os.path.join(HERE, "test_generic_gate.bin"), C)
    v = fx["vindex"]
    rows, _ = generic_test_circuit()
    ix = K.build_index(F, 5, rows)
    g = [C.srs_g(i) for i in range(32)]
    h = C.srs_h()
    basis = P.lagrange_basis(C, g, 5)
    commit = lambda col: P.commit_evaluations_non_hiding(C, basis, col, 5)
    assert [commit(ix["sigma"][i]) for i in range(7)] == v["sigma_comm"]
    assert [commit(ix["coefficients"][i]) for i in range(15)] == v["coefficients_comm"]
    assert P.mask_custom(C, h, commit(ix["generic_selector"]), [1]) == v["generic_comm"]
    for k in ("psm_comm", "complete_add_comm", "mul_comm", "emul_comm", "endomul_scalar_comm"):
        assert v[k] == [h]                          # the zero selector, masked with the blinder 1
    # the same through the oracle's NTT: interpolate (inverse transform), commit the coefficients over g
    for col, want in [(ix["sigma"][3], v["sigma_comm"][3]), (ix["coefficients"][8], v["coefficients_comm"][8])]:
        coeffs = P.ntt(F, col, 5, inverse=True)
        assert P.commit_non_hiding(C, g, coeffs, 1) == want
    # and the digest the transcript starts from
    vix = dict(v); vix["h"] = h
    mine = {"sigma_comm": [commit(ix["sigma"][i]) for i in range(7)], "coefficients_comm": [commit(ix["coefficients"][i]) for i in range(15)],
            "generic_comm": P.mask_custom(C, h, commit(ix["generic_selector"]), [1]), "psm_comm": [h], "complete_add_comm": [h], "mul_comm": [h],
            "emul_comm": [h], "endomul_scalar_comm": [h]}
    assert K.verifier_index_digest(C, mine) == K.verifier_index_digest(C, vix)
