"""Row a12: a COMPLETE Kimchi proof produced by the device pipeline (proof_systems_amd/prover.py -- witness ->
commitments -> iNTT -> z -> LDE -> constraint rows -> quotient with zero remainder -> t -> evaluations -> ft -> opening,
real Fiat-Shamir challenges from the library's native sponges) is ACCEPTED by the oracle's restatement of the reference
verifier (oracle/kimchi.py: transcript replayed with the oracle's own Poseidon, ft_eval0, ft_comm, SRS::verify); a
tampered proof is rejected; an unsatisfied circuit cannot be proved."""
import numpy as np
import pytest

from oracle import cref
from oracle import kimchi as K
from oracle import pasta as P
from oracle import views as V

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def _aff(c, xy, inf):
    return V.aff(c, xy, inf)


def _oracle_views(khip, ix, proof):
    """The device prover's index / proof as the plain-integer structures oracle/kimchi.py verifies."""
    return V.device_views(ix, proof)


def _bump(e, half, p):
    """an evaluation (chunks at zeta, chunks at zeta omega) with the first chunk of one half incremented"""
    a, b = list(e[0]), list(e[1])
    if half == 0:
        a[0] = (a[0] + 1) % p
    else:
        b[0] = (b[0] + 1) % p
    return (a, b)


def _verify(khip, ix, proof, seed=5):
    c, vix, pr = _oracle_views(khip, ix, proof)
    g_l = ix.srs.get_g()
    h = _aff(c, ix.h, False)
    ok = K.verify(c, vix, pr, None, h, P.StdRng(bytes([seed] * 32)), final_msm=V.final_msm_c(c, g_l, ix.size))
    return ok, (c, vix, pr)


@pytest.mark.parametrize("cid,logn", [(0, 7), (1, 7), (0, 12), (0, 16)])
def test_bench_circuit_proof_is_accepted_by_the_reference_verifier(khip, cid, logn):
    from proof_systems_amd import prover
    ix = prover.bench_circuit_index(cid, logn)
    F = prover.Fld(ix.fid)
    rows = (1 << logn) - 10
    wit = np.tile(F.limbs(1), (15, rows, 1))                          # kimchi/src/bench.rs:106
    t = {}
    proof = prover.create_proof(ix, wit, np.random.default_rng(11 + logn), timings=t)
    ok, (c, vix, pr) = _verify(khip, ix, proof)
    assert ok, "the oracle's verifier rejects the device prover's proof"
    # the challenges the native sponges produced are the ones the oracle's sponge derives from the same transcript
    ch = K.fiat_shamir(c, vix, pr, K.verifier_index_digest(c, vix))
    for k in ("beta", "gamma", "alpha", "zeta", "v", "u"):
        assert ch[k] == proof["challenges"][k], k
    if logn <= 12:
        bad = dict(proof, evals=dict(proof["evals"], z=_bump(proof["evals"]["z"], 1, F.p)))
        assert not _verify(khip, ix, bad)[0]                          # a tampered evaluation
        bad = dict(proof, ft_eval1=(proof["ft_eval1"] + 1) % F.p)
        assert not _verify(khip, ix, bad)[0]
    ix.free()


def test_copy_constraints_and_unsatisfied_witness(khip):
    """A circuit WITH copy constraints (random wiring between the first rows' cells, witness constant on the cycles): the
    proof verifies; breaking a gate makes the quotient division fail, breaking a copy constraint makes z end != 1."""
    from proof_systems_amd import prover
    cid, logn = 0, 8
    n = 1 << logn
    F = prover.Fld(khip.FP)
    c = P.CURVES[cid]
    rows = n - 10
    rnd = np.random.default_rng(3)
    # gates: w0 + w1 - w2 = 0 (coefficients 1, 1, -1) on every row; witness (a, b, a + b, ...) with wired columns 3..6 equal in pairs
    co = np.zeros((rows, 15, 4), dtype=np.uint64)
    co[:, 0] = F.limbs(1); co[:, 1] = F.limbs(1); co[:, 2] = F.limbs(F.p - 1)
    ix = prover.ProverIndex(cid, logn, co)
    sid = [pow(ix.omega, j, F.p) for j in range(n)]
    sigma = [[ix.shifts[i] * sid[j] % F.p for j in range(n)] for i in range(7)]
    wv = [[0] * rows for _ in range(15)]
    for j in range(rows):
        a, b = int(rnd.integers(1, 1 << 60)), int(rnd.integers(1, 1 << 60))
        wv[0][j], wv[1][j], wv[2][j] = a, b, (a + b) % F.p
    for j in range(0, rows - 1, 2):                                   # cells (3, j) and (4, j + 1) wired together
        v = int(rnd.integers(1, 1 << 60))
        wv[3][j] = v; wv[4][j + 1] = v
        sigma[3][j] = ix.shifts[4] * sid[j + 1] % F.p
        sigma[4][j + 1] = ix.shifts[3] * sid[j] % F.p
    ix.set_sigma(np.stack([F.limbs_many(col) for col in sigma]))
    wit = np.stack([F.limbs_many(col) for col in wv])
    proof = prover.create_proof(ix, wit, np.random.default_rng(5))
    assert _verify(khip, ix, proof)[0]
    bad = wit.copy(); bad[2, 17] = F.limbs(12345)                     # gate 17 no longer holds
    with pytest.raises(RuntimeError, match="vanishing"):
        prover.create_proof(ix, bad, np.random.default_rng(5))
    bad = wit.copy(); bad[3, 0] = F.limbs(777)                        # copy constraint (3, 0) = (4, 1) broken
    with pytest.raises(RuntimeError, match="accumulator"):
        prover.create_proof(ix, bad, np.random.default_rng(5))
    ix.free()


@pytest.mark.parametrize("cid", [0, 1])
def test_proof_with_lookups_is_accepted_by_the_reference_pinned_verifier(khip, cid):
    """A circuit in the shape of the reference's lookup tests (kimchi/src/tests/lookup.rs:38-170: Lookup gates into user tables with
    ids, here next to generic gates): the device prover commits the sorted columns and the aggregation, evaluates the lookup
    constraints on d8 with the powers alpha^24.., opens the extra polynomials -- and the oracle verifier, which accepts the reference's
    own stored Lookup-gate proofs (tests/test_reference_fixtures.py), accepts the proof; tampering and a value outside the table fail."""
    import random
    from proof_systems_amd import lookup as LK, prover
    logn = 9; n = 1 << logn
    rnd = random.Random(21)
    fid = khip.FP if cid == 0 else khip.FQ
    F = prover.Fld(fid)
    tables = [{"id": 0, "data": [list(range(40)), [0] + [rnd.randrange(F.p) for _ in range(39)]]},
              {"id": 3, "data": [list(range(25)), [rnd.randrange(F.p) for _ in range(25)]]}]
    ngen, nlook = 30, 200
    co = np.zeros((ngen, 15, 4), dtype=np.uint64)
    co[:, 0, :] = F.limbs(1); co[:, 4, :] = F.limbs(F.p - 7)          # generic rows: w0 - 7 = 0
    gates = ["Generic"] * ngen + ["Lookup"] * nlook + ["Zero"] * (n - 3 - ngen - nlook)
    rows = ngen + nlook
    wit = [[0] * rows for _ in range(15)]
    for r in range(ngen):
        wit[0][r] = 7
    for r in range(ngen, rows):
        t = tables[rnd.randrange(2)]
        wit[0][r] = t["id"]
        for i in range(3):
            e = rnd.randrange(len(t["data"][0]))
            wit[2 * i + 1][r], wit[2 * i + 2][r] = t["data"][0][e], t["data"][1][e]
    ix = prover.ProverIndex(cid, logn, co)
    ix.attach_lookup(LK.LookupIndex(fid, gates, tables, logn))
    w = np.stack([F.limbs_many(c) for c in wit])
    proof = prover.create_proof(ix, w, np.random.default_rng(8))
    ok, (c, vix, pr) = _verify(khip, ix, proof)
    assert ok
    assert len(pr["lookup"]["sorted"]) == 4 and vix["lookup_index"]["table_ids"] is not None
    nproof = prover.create_proof_native(ix, w, np.random.default_rng(8))           # kh_prove: Lookup gates into user tables with ids, both curves
    assert nproof["challenges"] == proof["challenges"] and V.device_views(ix, nproof)[2] == pr
    bad = dict(proof); be = dict(proof["evals"]); be["lookup_aggregation"] = _bump(be["lookup_aggregation"], 1, F.p); bad["evals"] = be
    assert not _verify(khip, ix, bad)[0]
    bad = dict(proof); be = dict(proof["evals"]); srt = list(be["lookup_sorted"]); srt[1] = _bump(srt[1], 0, F.p); be["lookup_sorted"] = srt; bad["evals"] = be
    assert not _verify(khip, ix, bad)[0]
    wit[2][ngen + 5] = (wit[2][ngen + 5] + 1) % F.p                     # a looked-up value that is not in its table
    with pytest.raises(ValueError):
        prover.create_proof(ix, np.stack([F.limbs_many(c) for c in wit]), np.random.default_rng(8))
    with pytest.raises(khip.KhError, match="not in the table"):
        prover.create_proof_native(ix, np.stack([F.limbs_many(c) for c in wit]), np.random.default_rng(8))


def test_proof_over_the_gate_library_is_accepted(khip):
    """One circuit with every always-present gate type -- generic rows, a Poseidon permutation (11 rows), complete additions incl. the
    doubling and inverse cases, a variable-base scalar multiplication, an endo scalar multiplication and an endo-scalar decomposition
    (witnesses by the reference's generators restated in oracle/gates.py): the device prover evaluates each gate's token program
    (proof_systems_amd/polish.py) on d8 next to the permutation rows, and the oracle verifier -- which accepts the reference's own
    stored proofs for each of these gate types -- accepts the proof; a broken gate row makes the prover fail at the zero-remainder check."""
    import random
    from proof_systems_amd import prover
    from test_gates import gate_rows, tables
    rnd = random.Random(77)
    F = prover.Fld(khip.FP)
    wrows, crows, types = [], [], []
    for r in range(6):
        wrows.append([5] + [0] * 14); crows.append([1, 0, 0, 0, F.p - 5] + [0] * 10); types.append("Generic")
    for name in ("Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar"):
        w, co, ngate = tables(name, rnd)
        live = set(gate_rows(name, ngate))
        for r, (wr, cr) in enumerate(zip(w, co)):
            wrows.append(list(wr)); crows.append(list(cr)); types.append(name if r in live else "Zero")
    rows = len(wrows)
    logn = 7
    assert rows + 3 <= 1 << logn
    co = np.stack([F.limbs_many(r) for r in crows])
    ix = prover.ProverIndex(khip.VESTA, logn, co, gate_types=types)
    wit = np.stack([F.limbs_many([wrows[r][c] for r in range(rows)]) for c in range(15)])
    proof = prover.create_proof(ix, wit, np.random.default_rng(12))
    ok, (c, vix, pr) = _verify(khip, ix, proof)
    assert ok
    for key in ("poseidon_selector", "complete_add_selector", "mul_selector", "emul_selector", "endomul_scalar_selector"):
        assert proof["evals"][key][0][0] != 0                              # every gate type is live in this proof
    bad = dict(proof); be = dict(proof["evals"]); w_ = list(be["w"]); w_[2] = _bump(w_[2], 0, F.p); be["w"] = w_; bad["evals"] = be
    assert not _verify(khip, ix, bad)[0]
    r0 = types.index("Poseidon") + 3
    wrows[r0][7] = (wrows[r0][7] + 1) % F.p
    wit = np.stack([F.limbs_many([wrows[r][c] for r in range(rows)]) for c in range(15)])
    with pytest.raises(RuntimeError):
        prover.create_proof(ix, wit, np.random.default_rng(12))
