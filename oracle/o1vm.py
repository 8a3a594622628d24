"""TEST INFRASTRUCTURE ONLY (see oracle/pasta.py): o1vm's second prover -- `o1vm/src/pickles/prover.rs:55-483` -- and its verifier
(`pickles/verifier.rs:65-278`) restated on the CPU, as the checker of proof_systems_amd/o1vm.py (a second caller of the
same commit / interpolate / d8 / open entry points).

The protocol (no permutation, no zero-knowledge rows, blinder 1 everywhere): interpolate every column (the `scratch_inverse` columns after a
batch inversion of their evaluations, :120-127; 72 dynamic selector columns from the per-row instruction index, :99-110), commit each
polynomial with `commit_custom(.., 1, [1])`, absorb, alpha = a RAW 128-bit challenge (:231), quotient = (sum_i alpha^i constraint_i) / Z_H
over d8 with the remainder asserted zero (:262-300), committed in 7 chunks with blinders 1, zeta (endo-mapped), every column and the
quotient's chunks evaluated at zeta / zeta omega, absorbed interleaved (:378-430), v, u, one IPA opening of all of them.

Parity: the reference holds no golden vector for this prover ("parity unpinned" by bytes); what pins this restatement is that its VERIFIER
half accepts its PROVER half (and rejects tampering), on top of building blocks that ARE pinned (commitments, sponges, SRS::open / verify:
oracle/pasta.py, oracle/poseidon.py).  Constraints are token programs over the column numbering of `get_all_columns`
(pickles/column_env.rs:47-68): scratch, scratch_inverse, lookup_state, instruction_counter, error, selectors."""
from typing import List, Sequence

from . import pasta as P
from . import poseidon as S
from . import prover as OPR

SCRATCH_SIZE, SCRATCH_SIZE_INVERSE, N_MIPS_SEL_COLS, DEGREE_QUOTIENT_POLYNOMIAL = 63, 12, 72, 7     # interpreters/mips/column.rs:40-52, pickles/mod.rs:27


def all_columns(F: P.Field, inputs, n: int) -> List[List[int]]:
    """d1 evaluations of every column in get_all_columns order; `inputs`: scratch (63 columns), scratch_inverse (12, BEFORE inversion),
    lookup_state, instruction_counter, error, selector (one index per row)."""
    p = F.p
    inv = lambda col: [F.inv(v) if v % p else 0 for v in col]                         # ark_ff::batch_inversion leaves zeros
    sel = [[1 if s % p == i else 0 for s in inputs["selector"]] for i in range(N_MIPS_SEL_COLS)]
    cols = list(inputs["scratch"]) + [inv(c) for c in inputs["scratch_inverse"]] + list(inputs["lookup_state"]) + [inputs["instruction_counter"], inputs["error"]] + sel
    assert len(inputs["scratch"]) == SCRATCH_SIZE and len(inputs["scratch_inverse"]) == SCRATCH_SIZE_INVERSE and all(len(c) == n for c in cols)
    return [[v % p for v in c] for c in cols]


def prove(curve: P.Curve, log2_n: int, srs: OPR.Srs, inputs, constraints, rng: P.StdRng):
    """constraints: [(tokens, constants)] -- P.polish_evaluate_rows programs over the columns above."""
    F = curve.scalar; p = F.p
    n = 1 << log2_n
    assert srs.size == n
    _, endo_r = P.endos(curve)
    omega = F.root_of_unity(log2_n)
    cols = all_columns(F, inputs, n)
    polys = [P.ntt(F, c, log2_n, inverse=True) for c in cols]
    comms = [srs.mask(srs.commit_non_hiding(q, 1), [1]) for q in polys]
    fq = S.DefaultFqSponge(curve)
    for c in comms:
        fq.absorb_g(c)
    alpha = fq.challenge()                                   # NOT mapped through the endomorphism (prover.rs:231)
    d8 = [P.ntt(F, q, log2_n + 3) for q in polys]
    t8 = [0] * (8 * n)
    ap = 1
    for toks, consts in constraints:
        rows = P.polish_evaluate_rows(F, toks, d8, consts, 8 * n, 1, 8)
        t8 = [(a + ap * b) % p for a, b in zip(t8, rows)]
        ap = ap * alpha % p
    f = P.ntt(F, t8, log2_n + 3, inverse=True)
    q = [0] * (7 * n)
    for i in range(7 * n - 1, -1, -1):
        q[i] = (f[i + n] + (q[i + n] if i + n < 7 * n else 0)) % p
    assert all((f[i] + q[i]) % p == 0 for i in range(n)), "The constraints are not satisfied since the remainder is not zero"
    q_comm = srs.mask(srs.commit_non_hiding(q, DEGREE_QUOTIENT_POLYNOMIAL), [1] * DEGREE_QUOTIENT_POLYNOMIAL)
    fq.absorb_g(q_comm)
    zeta = P.challenge_to_field(F, fq.challenge(), endo_r)
    zetaw = zeta * omega % p
    ev = [(OPR._horner(p, c, zeta), OPR._horner(p, c, zetaw)) for c in polys]
    q_ev = (OPR.evaluate_chunks(p, q, zeta, 7, n), OPR.evaluate_chunks(p, q, zetaw, 7, n))
    fq_before = fq.clone()
    fr = S.ArithmeticSponge(F)
    dg = fq.clone().challenge_fq()
    fr.absorb([dg if dg < p else 0])
    for a, b in ev:
        fr.absorb([a]); fr.absorb([b])
    for a, b in zip(*q_ev):
        fr.absorb([a]); fr.absorb([b])
    v = P.challenge_to_field(F, fr.squeeze() & ((1 << 128) - 1), endo_r)
    u = P.challenge_to_field(F, fr.squeeze() & ((1 << 128) - 1), endo_r)
    plnms = [(c, [1]) for c in polys] + [(q, [1] * 7)]
    opening = P.ipa_open(curve, [None] * n, srs.h, plnms, [zeta, zetaw], v, u, fq_before, rng, rounds_backend=lambda a, b, ub: OPR._Rounds(srs, a, b, ub))
    return {"commitments": comms, "zeta_evaluations": [e[0] for e in ev], "zeta_omega_evaluations": [e[1] for e in ev], "quotient_commitment": q_comm,
            "quotient_evaluations": q_ev, "opening": {k: opening[k] for k in ("lr", "delta", "z1", "z2", "sg")},
            "challenges": {"alpha": alpha, "zeta": zeta, "v": v, "u": u}}


def verify(curve: P.Curve, log2_n: int, srs: OPR.Srs, constraints, proof, rng: P.StdRng, final_msm=None) -> bool:
    """pickles/verifier.rs:65-278: the transcript replayed, sum_i alpha^i constraint_i on the stated evaluations (Curr = at zeta, Next = at
    zeta omega) against quotient(zeta) * (zeta^n - 1), then SRS::verify."""
    F = curve.scalar; p = F.p
    n = 1 << log2_n
    _, endo_r = P.endos(curve)
    omega = F.root_of_unity(log2_n)
    fq = S.DefaultFqSponge(curve)
    for c in proof["commitments"]:
        fq.absorb_g(c)
    alpha = fq.challenge()
    fq.absorb_g(proof["quotient_commitment"])
    zeta = P.challenge_to_field(F, fq.challenge(), endo_r)
    zetaw = zeta * omega % p
    fq_before = fq.clone()
    fr = S.ArithmeticSponge(F)
    dg = fq.clone().challenge_fq()
    fr.absorb([dg if dg < p else 0])
    ze, zwe = proof["zeta_evaluations"], proof["zeta_omega_evaluations"]
    for a, b in zip(ze, zwe):
        fr.absorb([a]); fr.absorb([b])
    qz, qzw = proof["quotient_evaluations"]
    for a, b in zip(qz, qzw):
        fr.absorb([a]); fr.absorb([b])
    num, ap = 0, 1
    two_rows = [[a, b] for a, b in zip(ze, zwe)]                         # row 0 = at zeta, "next" = at zeta omega
    for toks, consts in constraints:
        num = (num + ap * P.polish_evaluate_rows(F, toks, two_rows, consts, 1, 1, 1)[0]) % p
        ap = ap * alpha % p
    v = P.challenge_to_field(F, fr.squeeze() & ((1 << 128) - 1), endo_r)
    u = P.challenge_to_field(F, fr.squeeze() & ((1 << 128) - 1), endo_r)
    evaluations = [(c, [[a], [b]]) for c, a, b in zip(proof["commitments"], ze, zwe)] + [(proof["quotient_commitment"], [list(qz), list(qzw)])]
    zn = pow(zeta, n, p)
    quotient_zeta = OPR._horner(p, qz, zn)
    if quotient_zeta != num * F.inv((zn - 1) % p) % p:
        return False
    item = {"sponge": fq_before, "evaluation_points": [zeta, zetaw], "polyscale": v, "evalscale": u, "evaluations": evaluations, "opening": proof["opening"],
            "combined_inner_product": P.combined_inner_product(F, v, u, [e for _, e in evaluations])}
    g_terms, pts, sc = P.ipa_verify_terms(curve, n, srs.h, [item], rng)
    if final_msm is not None:
        return final_msm(g_terms, pts, sc)
    from . import views as V
    return V.final_msm_c(curve, srs.g, n)(g_terms, pts, sc)
