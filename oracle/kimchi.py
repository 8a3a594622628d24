"""TEST INFRASTRUCTURE ONLY (see oracle/pasta.py): the parts of kimchi AROUND the hot path that a complete proof needs,
restated in plain Python integers so that a proof produced by the device pipeline (proof_systems_amd/prover.py) can be
CHECKED the way the reference's verifier checks it -- for circuits of generic gates (the benchmark circuit of
kimchi/src/bench.rs:59-96), no lookups, no optional gates, no recursion, one chunk (SRS size = domain size).

  Shifts::new                    kimchi/src/circuits/polynomials/permutation.rs:140-199   (Blake2b-sampled coset shifts)
  constraint system / index     kimchi/src/circuits/constraints.rs:870-1010 (padding, zk_rows = 3, sid, sigma), gate.rs, wires.rs
  VerifierIndex::digest          kimchi/src/verifier_index.rs:405-500
  ProverProof::oracles           kimchi/src/verifier.rs:126-640   (Fiat-Shamir replay, ft_eval0)
  to_batch                       kimchi/src/verifier.rs:781-1000  (f_comm, ft_comm, the evaluation list)
  SRS::verify                    oracle/pasta.py ipa_verify_terms (pinned on the reference's opening-proof bytes)
  generic gate                   kimchi/src/circuits/polynomials/generic.rs:83-120, argument.rs:201-214
  alpha powers                   kimchi/src/linearization.rs:43-58,167-171: gates 0..20, permutation 21..23

PINNED on the reference's stored proofs (kimchi/src/tests/fixtures/*.bin, copies in tests/golden/ref_fixtures/): `verify` accepts
eight proofs the reference's own prover produced (generic gates with / without public inputs, Poseidon, CompleteAdd, VarBaseMul,
EndoMul, EndoMulScalar) and rejects tampered ones, and `build_index` + commitments reproduce the reference's serialised verifier
index byte for byte (tests/test_reference_fixtures.py).  Byte parity of a whole PROOF is not claimed (Rust RNG stream)."""
import hashlib
from typing import List, Optional, Sequence

from . import pasta as P
from . import poseidon as S

COLUMNS, PERMUTS, ZK_ROWS = 15, 7, 3
ALPHA_PERM0 = 21           # VarbaseMul::CONSTRAINTS = 21 powers are registered for the gates first (linearization.rs:56-58)


# ---------------------------------------------------------------------------------------------------- index
def sample_shifts(F: P.Field, log2_n: int) -> List[int]:
    """Shifts::new: shift_0 = 1, the others quadratic non-residues outside the domain sampled from Blake2b512(counter)."""
    n = 1 << log2_n
    counter = [7]

    def sample():
        while True:
            counter[0] += 1
            d = hashlib.blake2b(counter[0].to_bytes(4, "big"), digest_size=64).digest()
            s = int.from_bytes(d[:31], "little")                       # F::from_random_bytes(&h[..31])
            if pow(s, (F.p - 1) // 2, F.p) == F.p - 1 and pow(s, n, F.p) != 1:
                return s
    shifts = [1]
    for _ in range(1, PERMUTS):
        s = sample()
        while s in shifts:
            s = sample()
        shifts.append(s)
    return shifts


def generic_const_row(F: P.Field, c: int) -> List[int]:
    """CircuitGate::create_generic_gadget(wires, GenericGateSpec::Const(c), None): coefficients of one row."""
    co = [0] * COLUMNS
    co[0] = 1; co[4] = (-c) % F.p
    return co


def build_index(F: P.Field, log2_n: int, gate_coeffs: Sequence[Sequence[int]], wiring=None):
    """Constraint system of `len(gate_coeffs)` generic gates on a domain of 2^log2_n rows: the remaining rows are Zero
    gates wired to themselves.  wiring[(col, row)] = (col', row') overrides Wire::for_row (the identity)."""
    n = 1 << log2_n
    assert len(gate_coeffs) + ZK_ROWS <= n
    omega = F.root_of_unity(log2_n)
    sid = [1] * n
    for j in range(1, n):
        sid[j] = sid[j - 1] * omega % F.p
    shifts = sample_shifts(F, log2_n)
    coeffs = [[0] * n for _ in range(COLUMNS)]
    sel = [0] * n
    for r, co in enumerate(gate_coeffs):
        sel[r] = 1
        for c in range(COLUMNS):
            coeffs[c][r] = co[c] % F.p
    sigma = [[shifts[c] * sid[r] % F.p for r in range(n)] for c in range(PERMUTS)]
    if wiring:
        for (c, r), (c2, r2) in wiring.items():
            sigma[c][r] = shifts[c2] * sid[r2] % F.p
    return {"F": F, "log2_n": log2_n, "n": n, "omega": omega, "sid": sid, "shifts": shifts, "coefficients": coeffs, "generic_selector": sel,
            "sigma": sigma, "zk_rows": ZK_ROWS, "gates": len(gate_coeffs)}


def eval_permutation_vanishing_polynomial(ix, x: int) -> int:
    F = ix["F"]; n = ix["n"]
    t = pow(ix["omega"], n - ix.get("zk_rows", ZK_ROWS), F.p)
    return (x - t) * (x - t * ix["omega"]) % F.p * (x - pow(ix["omega"], n - 1, F.p)) % F.p


def absorb_commitment(sponge, chunks):
    sponge.absorb_g(chunks)


def verifier_index_digest(curve: P.Curve, vix) -> int:
    sp = S.DefaultFqSponge(curve)
    for c in vix["sigma_comm"]:
        absorb_commitment(sp, c)
    for c in vix["coefficients_comm"]:
        absorb_commitment(sp, c)
    for k in ("generic_comm", "psm_comm", "complete_add_comm", "mul_comm", "emul_comm", "endomul_scalar_comm"):
        absorb_commitment(sp, vix[k])
    opt = vix.get("optional_comms") or [None] * 6           # struct order: range_check0, range_check1, foreign_field_add, foreign_field_mul, xor, rot
    for k in (0, 1, 3, 2, 4, 5):                            # the digest absorbs foreign_field_mul BEFORE foreign_field_add (verifier_index.rs:466-480)
        if opt[k] is not None:
            absorb_commitment(sp, opt[k])
    li = vix.get("lookup_index")
    if li:                                                 # verifier_index.rs:482-525: table columns, table ids, (runtime selector), pattern selectors
        for c in li["lookup_table"]:
            absorb_commitment(sp, c)
        if li["table_ids"] is not None:
            absorb_commitment(sp, li["table_ids"])
        if li.get("runtime_tables_selector") is not None:
            absorb_commitment(sp, li["runtime_tables_selector"])
        for q in LOOKUP_PATTERN_ORDER:
            if li["lookup_selectors"].get(q) is not None:
                absorb_commitment(sp, li["lookup_selectors"][q])
    return sp.challenge_fq()                               # digest_fq


# ---------------------------------------------------------------------------------------------------- verifier
def public_commitment(curve: P.Curve, h, lagrange_commitments, public: Sequence[int]):
    """verifier.rs:834-858: the commitment to the (negated) public-input polynomial over the Lagrange basis, masked with the
    blinder 1; `lagrange_commitments[i]` = the commitment to L_i (one chunk).  Empty public input: the blinding commitment."""
    if not public:
        return [h]
    F = curve.scalar
    acc = None
    for L, x in zip(lagrange_commitments, public):
        acc = curve.add(acc, curve.mul(L, (-x) % F.p))
    return [curve.add(acc, h)]


def public_evaluations(F: P.Field, n: int, omega: int, public: Sequence[int], zeta: int):
    """verifier.rs:336-386: (p(zeta), p(zeta omega)) of the negated public-input polynomial -sum_i pub_i L_i, from the inputs."""
    out = []
    for x in (zeta, zeta * omega % F.p):
        zh = (pow(x, n, F.p) - 1) % F.p
        acc = 0
        for i, v in enumerate(public):
            wi = pow(omega, i, F.p)
            acc = (acc - v * wi % F.p * F.inv((x - wi) % F.p)) % F.p
        out.append(acc * zh % F.p * F.inv(n % F.p) % F.p)
    return tuple(out)


def generic_constant_term(F: P.Field, ev, alpha: int) -> int:
    """index(Generic) * (alpha^0 c1 + alpha^1 c2) at zeta from the proof's evaluations (the only non-zero part of
    linearization.constant_term for a circuit whose other selectors are the zero polynomial)."""
    w = [e[0] for e in ev["w"]]; co = [e[0] for e in ev["coefficients"]]
    c1 = (co[0] * w[0] + co[1] * w[1] + co[2] * w[2] + co[3] * w[0] * w[1] + co[4]) % F.p
    c2 = (co[5] * w[3] + co[6] * w[4] + co[7] * w[5] + co[8] * w[3] * w[4] + co[9]) % F.p
    return ev["generic_selector"][0] * (c1 + alpha * c2) % F.p


GATE_SELECTORS = (("Poseidon", "poseidon_selector"), ("CompleteAdd", "complete_add_selector"), ("VarBaseMul", "mul_selector"),
                  ("EndoMul", "emul_selector"), ("EndoMulScalar", "endomul_scalar_selector"))
# the optional gates, in the order of ProofEvaluations / the Fr-sponge / the opening (proof.rs:95-106, plonk_sponge.rs:103-122,
# verifier.rs:1003-1038); evals["optional_gate_selectors"][k] is None for a gate type the circuit does not use
OPTIONAL_GATES = ("RangeCheck0", "RangeCheck1", "ForeignFieldAdd", "ForeignFieldMul", "Xor16", "Rot64")


def optional_selectors(ev):
    return list(ev.get("optional_gate_selectors") or [None] * 6)


def gate_library_constant_term(curve: P.Curve, ev, alpha: int) -> int:
    """sum over the five always-present gate types of selector(zeta) * (sum_i alpha^i constraint_i) on the proof's evaluations:
    curr = w(zeta), next = w(zeta omega), coefficients(zeta) -- the rest of linearization.constant_term (linearization.rs:43-100;
    every gate's constraints start at alpha^0).  The row machines are oracle/gates.py's."""
    from . import gates as G
    F = curve.scalar
    total = 0
    live = [(name, ev[key][0]) for name, key in GATE_SELECTORS if ev[key][0] % F.p]
    live += [(name, e[0]) for name, e in zip(OPTIONAL_GATES, optional_selectors(ev)) if e is not None]
    if not live:
        return 0
    curr = [e[0] for e in ev["w"]]; nxt = [e[1] for e in ev["w"]]; co = [e[0] for e in ev["coefficients"]]
    mds = S.params("fp" if F is P.Fp else "fq")["mds"]
    endo = P.endos(P.PALLAS if curve is P.VESTA else P.VESTA)[0]        # VerifierIndex::endo = endos::<G::OtherCurve>().0 (an element of G's scalar field)
    for name, sel in live:
        total = (total + sel * G.combined_row(F, name, curr, nxt, co, alpha, mds=mds, endo=endo)) % F.p
    return total


def perm_scalars(F: P.Field, ev, beta: int, gamma: int, alpha0: int, zkp_zeta: int) -> int:
    r = ev["z"][1] * beta % F.p * alpha0 % F.p * zkp_zeta % F.p
    for w, s in zip(ev["w"], ev["s"]):
        r = r * ((gamma + beta * s[0] + w[0]) % F.p) % F.p
    return (-r) % F.p


EVAL_ORDER = ["z", "generic_selector", "poseidon_selector", "complete_add_selector", "mul_selector", "emul_selector", "endomul_scalar_selector"]


def columns_in_opening_order(ev):
    """z, the six selectors, w x 15, coefficients x 15, s x 6: FrSponge::absorb_evaluations (plonk_sponge.rs:92-155) and
    the verifier's evaluation list (verifier.rs:988-1010) use the same order."""
    out = [ev[k] for k in EVAL_ORDER]
    out += list(ev["w"]) + list(ev["coefficients"]) + list(ev["s"])
    out += [e for e in optional_selectors(ev) if e is not None]
    return out


LOOKUP_PATTERN_ORDER = ("Xor", "Lookup", "RangeCheck", "ForeignFieldMul")
ALPHA_LOOKUP0 = ALPHA_PERM0 + 3        # linearization.rs:170-186: the lookup constraints are registered after the permutation's three powers


def lookup_evaluations_in_sponge_order(ev, li):
    """FrSponge::absorb_evaluations (plonk_sponge.rs:127-155): aggregation, table, sorted..., the pattern selectors."""
    if not li:
        return []
    rt = [ev[k] for k in ("runtime_lookup_table", "runtime_lookup_table_selector") if ev.get(k) is not None]
    return [ev["lookup_aggregation"], ev["lookup_table"]] + list(ev["lookup_sorted"]) + rt + [ev["lookup_selectors"][q] for q in LOOKUP_PATTERN_ORDER if q in ev["lookup_selectors"]]


def lookup_constant_term(F: P.Field, vix, ev, ch, zeta: int) -> int:
    """sum_k alpha^(24 + k) * lookup constraint_k on the proof's evaluations: the lookup part of linearization.constant_term.
    The constraints are oracle/lookup.py's (constraints.rs:378-673); the table-id combiner is joint_combiner^max_joint_size
    unconditionally there (constraints.rs:424-440), cells of row Curr / Next are the evaluations at zeta / zeta omega."""
    from . import lookup as L
    li = vix["lookup_index"]
    n, omega, zk = vix["n"], vix["omega"], vix["zk_rows"]
    jc = ch["joint_combiner"]

    class Shim:                                                         # what constraint_values reads from a LookupCS
        p = F.p

        class info:
            patterns = [q for q in LOOKUP_PATTERN_ORDER if q in li["patterns"]]
            max_per_row = li["max_per_row"]

        @staticmethod
        def constraint_combiners(j):
            return j % F.p, pow(j, li["max_joint_size"], F.p)

        @staticmethod
        def dummy_value(j):
            return 0
    cols = {"w": ev["w"], "sorted": ev["lookup_sorted"], "aggreg": [ev["lookup_aggregation"]], "table": [ev["lookup_table"]]}

    def cell(kind, idx, row):
        if kind == "selector":
            return ev["lookup_selectors"][idx][0]
        return cols[kind][idx][row]
    atoms = {"vanish": L.vanishes_on_last_n_rows(F.p, omega, n, zk + 1, zeta),
             "l0": L.unnormalized_lagrange_basis(F.p, omega, n, 0, zeta),
             "lfinal": L.unnormalized_lagrange_basis(F.p, omega, n, -(zk + 1), zeta)}
    vals = L.constraint_values(Shim, jc, ch["beta"], ch["gamma"], cell, atoms)
    if li.get("uses_runtime_tables"):                                   # runtime_tables::constraints (runtime_tables.rs:59-66), after the padding to four
        vals.append(ev["runtime_lookup_table"][0] * ev["runtime_lookup_table_selector"][0] % F.p)
    return sum(pow(ch["alpha"], ALPHA_LOOKUP0 + k, F.p) * v for k, v in enumerate(vals)) % F.p


def lookup_table_commitment(curve: P.Curve, li, jc: int, runtime=None):
    """combine_table (lookup/tables/mod.rs:164-199): sum_i jc^i * column_i + jc^max_joint_size * table_ids (+ jc * runtime), per chunk."""
    F = curve.scalar
    nch = len(li["lookup_table"][0])
    out = []
    for k in range(nch):
        acc, j = None, 1
        for c in li["lookup_table"]:
            if c[k] is not None:
                acc = curve.add(acc, curve.mul(c[k], j))
            j = j * jc % F.p
        if li["table_ids"] is not None and li["table_ids"][k] is not None:
            acc = curve.add(acc, curve.mul(li["table_ids"][k], pow(jc, li["max_joint_size"], F.p)))
        if runtime is not None and runtime[k] is not None:
            acc = curve.add(acc, curve.mul(runtime[k], jc))
        out.append(acc)
    return out


def _chunks(e):
    """an evaluation as (chunks at zeta, chunks at zeta omega): plain pairs of integers are one-chunk evaluations"""
    if e is None:
        return None
    a, b = e
    return (list(a) if isinstance(a, (list, tuple)) else [a], list(b) if isinstance(b, (list, tuple)) else [b])


def normalize_evals(ev):
    """ProofEvaluations<PointEvaluations<Vec<F>>>: every entry as chunk lists."""
    out = dict(ev)
    for k in EVAL_ORDER + ["public", "lookup_aggregation", "lookup_table", "runtime_lookup_table", "runtime_lookup_table_selector"]:
        if k in ev:
            out[k] = _chunks(ev[k])
    for k in ("w", "coefficients", "s", "lookup_sorted"):
        out[k] = [_chunks(e) for e in ev.get(k, [])]
    out["optional_gate_selectors"] = [_chunks(e) for e in optional_selectors(ev)]
    out["lookup_selectors"] = {q: _chunks(e) for q, e in (ev.get("lookup_selectors") or {}).items()}
    return out


def combine_evals(F: P.Field, evn, zeta_srs: int, zetaw_srs: int):
    """ProofEvaluations::combine (proof.rs:430-470): sum_c chunk_c * (zeta^max_poly_size)^c -- the value of the whole polynomial."""
    def hz(e):
        if e is None:
            return None
        a = b = 0
        for c in reversed(e[0]):
            a = (a * zeta_srs + c) % F.p
        for c in reversed(e[1]):
            b = (b * zetaw_srs + c) % F.p
        return (a, b)
    out = dict(evn)
    for k, v in evn.items():
        if k == "lookup_selectors":
            out[k] = {q: hz(e) for q, e in v.items()}
        elif isinstance(v, list):
            out[k] = [hz(e) for e in v]
        else:
            out[k] = hz(v)
    return out


def prev_challenge_evals(F: P.Field, chals, max_poly_size: int, points, powers):
    """RecursionChallenge::evals (proof.rs:455-494): b_poly at both points, split into two chunks when 2^len(chals) > max_poly_size."""
    b_len = 1 << len(chals)
    out = []
    coeffs = None
    for x, pw in zip(points, powers):
        full = P.b_poly(F, chals, x)
        if max_poly_size == b_len:
            out.append([full]); continue
        coeffs = coeffs or P.b_poly_coefficients(F, chals)
        diff, acc = 0, 1
        for j in range(max_poly_size, b_len):
            diff = (diff + acc * coeffs[j]) % F.p
            acc = acc * x % F.p
        out.append([(full - diff * pw) % F.p, diff])
    return out


def default_public_comm(vix):
    """verifier.rs:844-846: an empty public input commits to the blinding commitment, once per chunk."""
    n, size = vix["n"], vix.get("max_poly_size", vix["n"])
    return [vix["h"]] * (1 if n < size else n // size)


def fiat_shamir(curve: P.Curve, vix, proof, digest: int):
    """verifier.rs:160-420: returns the challenges and the Fq-sponge as SRS::verify needs it."""
    F = curve.scalar
    _, endo_r = P.endos(curve)
    fq = S.DefaultFqSponge(curve)
    fq.absorb_fq([digest])
    prev = proof.get("prev_challenges") or []
    for _chals, comm in prev:
        absorb_commitment(fq, comm)
    absorb_commitment(fq, vix.get("public_comm") or default_public_comm(vix))
    for c in proof["w_comm"]:
        absorb_commitment(fq, c)
    li = vix.get("lookup_index")
    joint_combiner = None
    if li:                                                              # verifier.rs:179-230
        if li.get("runtime_tables_selector") is not None:
            absorb_commitment(fq, proof["lookup"]["runtime"])
        joint_combiner = P.challenge_to_field(F, fq.challenge() if li["joint_lookup_used"] else 0, endo_r)
        for c in proof["lookup"]["sorted"]:
            absorb_commitment(fq, c)
    beta = fq.challenge(); gamma = fq.challenge()
    if li:
        absorb_commitment(fq, proof["lookup"]["aggreg"])
    absorb_commitment(fq, proof["z_comm"])
    alpha = P.challenge_to_field(F, fq.challenge(), endo_r)
    n = vix["n"]
    size = vix.get("max_poly_size", n)
    assert len(proof["t_comm"]) <= 7 * max(1, n // size)
    absorb_commitment(fq, proof["t_comm"])
    zeta = P.challenge_to_field(F, fq.challenge(), endo_r)
    dg = fq.clone().challenge_fq()
    dg = dg if dg < F.p else 0                                          # FqSponge::digest
    fr = S.ArithmeticSponge(F)
    fr.absorb([dg])
    pd = S.ArithmeticSponge(F)
    for chals, _comm in prev:
        pd.absorb(list(chals))
    fr.absorb([pd.squeeze()])                                           # prev_challenge_digest
    ev = normalize_evals(proof["evals"])
    fr.absorb([proof["ft_eval1"]])
    fr.absorb(ev["public"][0]); fr.absorb(ev["public"][1])
    for col in columns_in_opening_order(ev) + lookup_evaluations_in_sponge_order(ev, li):
        fr.absorb(col[0]); fr.absorb(col[1])

    def fr_challenge():                                                 # DefaultFrSponge::challenge: 128 bits of one squeeze (the
        x = fr.squeeze()                                                # buffer is emptied by every absorb; two limbs per squeeze)
        return x & ((1 << 128) - 1)
    # v and u come from consecutive challenge() calls: last_squeezed holds exactly the two limbs of one squeeze
    v = P.challenge_to_field(F, fr_challenge(), endo_r)
    u = P.challenge_to_field(F, fr_challenge(), endo_r)
    return {"beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v, "u": u, "fq_sponge": fq, "joint_combiner": joint_combiner}


def verify(curve: P.Curve, vix, proof, g, h, rng, final_msm=None) -> bool:
    """kimchi::verifier::verify (verifier.rs:781-1200 + SRS::verify).  vix: n, log2_n, omega, shifts, h, max_poly_size, zk_rows and the
    index commitments; proof: w_comm, z_comm, t_comm (chunk lists of affine points / None), evals (pairs of integers or of chunk
    lists), ft_eval1, opening, optionally lookup and prev_challenges [(chals, comm chunks)]."""
    F = curve.scalar
    n = vix["n"]
    ch = fiat_shamir(curve, vix, proof, verifier_index_digest(curve, vix))
    beta, gamma, alpha, zeta, v, u = ch["beta"], ch["gamma"], ch["alpha"], ch["zeta"], ch["v"], ch["u"]
    omega = vix["omega"]
    zetaw = zeta * omega % F.p
    zeta1 = pow(zeta, n, F.p)
    srs_len = vix.get("max_poly_size", n)                               # chunk length = SRS size (verifier.rs:795, zeta_to_srs_len)
    zeta_srs = pow(zeta, srs_len, F.p); zetaw_srs = pow(zetaw, srs_len, F.p)
    evc = normalize_evals(proof["evals"])                               # chunked, as absorbed and opened
    ev = combine_evals(F, evc, zeta_srs, zetaw_srs)                     # combined, as the constraints see them
    zk_rows = vix.get("zk_rows", ZK_ROWS)
    alphas = [pow(alpha, ALPHA_PERM0 + i, F.p) for i in range(3)]
    zkp = eval_permutation_vanishing_polynomial(vix, zeta)
    # ---- ft_eval0 (verifier.rs:412-490)
    ft0 = (ev["w"][PERMUTS - 1][0] + gamma) * ev["z"][1] % F.p * alphas[0] % F.p * zkp % F.p
    for w, s in zip(ev["w"], ev["s"]):
        ft0 = ft0 * ((beta * s[0] + w[0] + gamma) % F.p) % F.p
    ft0 = (ft0 - ev["public"][0]) % F.p
    t = alphas[0] * zkp % F.p * ev["z"][0] % F.p
    for w, sh in zip(ev["w"], vix["shifts"]):
        t = t * ((gamma + beta * zeta % F.p * sh + w[0]) % F.p) % F.p
    ft0 = (ft0 - t) % F.p
    zeta1m1 = (zeta1 - 1) % F.p
    w_zk = pow(omega, n - zk_rows, F.p)                                 # index.w()
    num = (zeta1m1 * alphas[1] % F.p * (zeta - w_zk) + zeta1m1 * alphas[2] % F.p * (zeta - 1)) % F.p * ((1 - ev["z"][0]) % F.p) % F.p
    den = (zeta - w_zk) * (zeta - 1) % F.p
    ft0 = (ft0 + num * F.inv(den)) % F.p
    ft0 = (ft0 - generic_constant_term(F, ev, alpha)) % F.p
    ft0 = (ft0 - gate_library_constant_term(curve, ev, alpha)) % F.p
    li = vix.get("lookup_index")
    if li:
        ft0 = (ft0 - lookup_constant_term(F, vix, ev, ch, zeta)) % F.p
    # ---- commitments: f_comm = perm_scalar * sigma_comm[6] chunk-combined; ft_comm = f_comm - (zeta^n - 1) * sum_i zeta^(srs_len i) t_comm[i]
    scal = perm_scalars(F, ev, beta, gamma, alphas[0], zkp)

    def chunk_commitment(chunks, pw0):                                  # PolyComm::chunk_commitment (commitment.rs:188-205)
        acc, pw = None, 1
        for c in chunks:
            if c is not None:
                acc = curve.add(acc, curve.mul(c, pw))
            pw = pw * pw0 % F.p
        return acc
    f_comm = chunk_commitment([curve.mul(c, scal) if c is not None else None for c in vix["sigma_comm"][PERMUTS - 1]], zeta_srs)
    t_chunk = chunk_commitment(proof["t_comm"], zeta_srs)
    neg = curve.mul(t_chunk, (-zeta1m1) % F.p) if t_chunk is not None else None
    ft_comm = curve.add(f_comm, neg)
    # ---- the evaluation list: previous challenges, public, ft, then the columns in opening order
    evaluations = []
    for chals, comm in (proof.get("prev_challenges") or []):
        evaluations.append((list(comm), prev_challenge_evals(F, chals, srs_len, [zeta, zetaw], [zeta_srs, zetaw_srs])))
    evaluations += [(vix.get("public_comm") or default_public_comm(vix), [evc["public"][0], evc["public"][1]]), ([ft_comm], [[ft0], [proof["ft_eval1"]]])]
    comms = [proof["z_comm"], vix["generic_comm"], vix["psm_comm"], vix["complete_add_comm"], vix["mul_comm"], vix["emul_comm"], vix["endomul_scalar_comm"]]
    comms += list(proof["w_comm"]) + list(vix["coefficients_comm"]) + list(vix["sigma_comm"][:PERMUTS - 1])
    comms += [c for c in (vix.get("optional_comms") or []) if c is not None]
    assert len(comms) == len(columns_in_opening_order(evc)), "optional gate commitments / evaluations mismatch"
    for c, e in zip(comms, columns_in_opening_order(evc)):
        evaluations.append((c, [e[0], e[1]]))
    if li:                                                              # verifier.rs:1034-1175: sorted..., aggregation, the combined table, the pattern selectors
        lk = [(c, e) for c, e in zip(proof["lookup"]["sorted"], evc["lookup_sorted"])] + [(proof["lookup"]["aggreg"], evc["lookup_aggregation"])]
        lk.append((lookup_table_commitment(curve, li, ch["joint_combiner"], proof["lookup"].get("runtime")), evc["lookup_table"]))
        if li.get("runtime_tables_selector") is not None:
            lk.append((proof["lookup"]["runtime"], evc["runtime_lookup_table"]))
            lk.append((li["runtime_tables_selector"], evc["runtime_lookup_table_selector"]))
        lk += [(li["lookup_selectors"][q], evc["lookup_selectors"][q]) for q in LOOKUP_PATTERN_ORDER if li["lookup_selectors"].get(q) is not None]
        for c, e in lk:
            evaluations.append((c, [e[0], e[1]]))
    item = {"sponge": ch["fq_sponge"], "evaluation_points": [zeta, zetaw], "polyscale": v, "evalscale": u, "evaluations": evaluations,
            "opening": proof["opening"], "combined_inner_product": P.combined_inner_product(F, v, u, [e for _, e in evaluations])}
    if final_msm is None:
        return P.ipa_verify(curve, g, h, [item], rng)
    g_terms, pts, sc = P.ipa_verify_terms(curve, srs_len, h, [item], rng)
    return final_msm(g_terms, pts, sc)
