"""o1vm's second prover on the device -- another caller of the same hot path.

`o1vm/src/pickles/prover.rs:55-483` proves MIPS-interpreter traces with a Plonk-ish protocol of its own: ~150 witness columns (63 scratch,
12 scratch-inverse, lookup state, instruction counter, error, 72 dynamic selectors), no permutation argument, no zero-knowledge rows, every
commitment blinded with 1.  Its data-parallel steps are exactly the entry points of this library: interpolate every column (kh_ntt_dev, the
inverse columns after kh_batch_inversion_dev), commit them over the monomial basis (one batched MSM for all columns), 8x extension
(kh_lde_dev), the caller's constraints as token programs over d8 (kh_expr_evaluations_dev, powers of the RAW alpha challenge), iNTT(8n),
division by Z_H with the remainder asserted zero, a 7-chunk quotient commitment, chunked evaluations, one IPA opening (kh_ipa_open).  The
constraints themselves (the MIPS interpreter's, thousands of lines of Rust `Expr`) are the caller's, lowered with `Expr::to_polish()`: here they
arrive as token programs over the column numbering of `get_all_columns` (pickles/column_env.rs:47-68).  Product code: never imports the oracle;
checked against oracle/o1vm.py (its verifier half restates pickles/verifier.rs) in tests/test_gpu_o1vm.py."""
from __future__ import annotations

import numpy as np

from . import khip
from . import polish as OP
from .prover import Fld, scalar_challenge

SCRATCH_SIZE, SCRATCH_SIZE_INVERSE, N_MIPS_SEL_COLS, DEGREE_QUOTIENT_POLYNOMIAL = 63, 12, 72, 7     # interpreters/mips/column.rs:40-52, pickles/mod.rs:27


def prove(curve: int, log2_n: int, srs, inputs, constraints, rng, check: bool = True):
    """inputs: scratch (63, n, 4), scratch_inverse (12, n, 4: values to be inverted), lookup_state (L, n, 4), instruction_counter (n, 4),
    error (n, 4) -- Montgomery limbs -- and selector: n integers (the instruction index of each row).  constraints: [(tokens, constants as
    integers)].  rng: the caller's generator (the opening's blinders).  Returns the proof (pickles/proof.rs:33-43) as limb arrays / integers."""
    fid = khip.FP if curve == khip.VESTA else khip.FQ
    F = Fld(fid)
    n = 1 << log2_n
    NB, N8 = n * 32, 8 * n * 32
    assert srs.n == n, "o1vm proves over an SRS of the domain's size"
    lk = np.asarray(inputs["lookup_state"], dtype=np.uint64).reshape(-1, n, 4)
    ncol = SCRATCH_SIZE + SCRATCH_SIZE_INVERSE + lk.shape[0] + 2 + N_MIPS_SEL_COLS
    one = F.limbs(1)
    # ---- every column on d1, in get_all_columns order; the inverse columns inverted in place (zeros stay: ark_ff::batch_inversion)
    ev = khip.DevBuf(ncol * NB)
    ev.upload_at(0, np.asarray(inputs["scratch"], dtype=np.uint64).reshape(SCRATCH_SIZE, n, 4))
    o_inv = SCRATCH_SIZE * NB
    ev.upload_at(o_inv, np.asarray(inputs["scratch_inverse"], dtype=np.uint64).reshape(SCRATCH_SIZE_INVERSE, n, 4))
    khip.batch_inversion_dev(fid, ev.view(o_inv), SCRATCH_SIZE_INVERSE * n)
    o = o_inv + SCRATCH_SIZE_INVERSE * NB
    if lk.shape[0]:
        ev.upload_at(o, lk); o += lk.shape[0] * NB
    ev.upload_at(o, np.asarray(inputs["instruction_counter"], dtype=np.uint64).reshape(n, 4)); o += NB
    ev.upload_at(o, np.asarray(inputs["error"], dtype=np.uint64).reshape(n, 4)); o += NB
    sel = np.zeros((N_MIPS_SEL_COLS, n, 4), dtype=np.uint64)
    idx = np.asarray(inputs["selector"], dtype=np.int64)
    rows = np.nonzero((idx >= 0) & (idx < N_MIPS_SEL_COLS))[0]
    sel[idx[rows], rows] = one                               # selector i is 1 on the rows whose instruction is i (prover.rs:99-110)
    ev.upload_at(o, sel)
    # ---- interpolate, commit (commit_custom(poly, 1, [1]): one batched MSM over g, then + h), extend
    cf = khip.DevBuf(ncol * NB)
    khip.dev_copy(cf.ptr, ev.ptr, ncol * NB)
    khip.ntt_dev(fid, cf, log2_n, True, ncol)
    com, inf = srs.msm_batch_dev(cf.ptr, n, ncol)
    cxy, cinf = srs.mask_custom(com, inf, F.limbs_many([1] * ncol))
    e8 = khip.DevBuf(ncol * N8)
    khip.lde_dev(fid, cf, log2_n, 3, e8, ncol)
    fq = khip.Sponge(khip.Sponge.FQ, curve)
    for i in range(ncol):
        fq.absorb_g(cxy[i:i + 1], cinf[i:i + 1])
    alpha = fq.challenge()                                  # the raw 128-bit challenge IS alpha (prover.rs:231)
    # ---- quotient: sum_i alpha^i constraint_i over d8, interpolated, / Z_H
    t8 = khip.DevBuf(N8)
    bufs = [e8.view(c * N8) for c in range(ncol)]
    ap = 1
    for k, (toks, consts) in enumerate(constraints):
        cs = [int(c) % F.p for c in consts] + [ap]
        khip.expr_evaluations_dev(fid, list(toks) + [(OP.TOK_CONST, len(cs) - 1), (OP.TOK_MUL, 0)], bufs, [8 * n] * ncol, F.limbs_many(cs), 8 * n, t8,
                                  stride=1, next_shift=8, accumulate=k > 0)
        ap = ap * alpha % F.p
    khip.ntt_dev(fid, t8, log2_n + 3, True, 1)
    quot = khip.DevBuf(7 * NB); rem = khip.DevBuf(NB)
    khip.divide_by_vanishing_poly_dev(fid, t8, 8 * n, log2_n, quot, rem)
    if check and rem.download((n, 4)).any():
        raise RuntimeError("The constraints are not satisfied since the remainder is not zero (pickles/prover.rs:283-291)")
    qcom, qinf = srs.msm_batch_dev(quot.ptr, n, DEGREE_QUOTIENT_POLYNOMIAL)
    qxy, qci = srs.mask_custom(qcom, qinf, F.limbs_many([1] * DEGREE_QUOTIENT_POLYNOMIAL))
    fq.absorb_g(qxy, qci)
    zeta = scalar_challenge(curve, F, fq.challenge())
    omega = F.value(khip.domain_generator(fid, log2_n))
    zetaw = zeta * omega % F.p
    fq_before = fq.clone()
    # ---- evaluations: every column (one chunk), the quotient in 7 chunks of n
    pts = F.limbs_many([zeta, zetaw])
    polys = [cf.view(c * NB) for c in range(ncol)] + [quot]
    evl = khip.evaluate_chunks_batch_dev(fid, polys, [n] * ncol + [7 * n], [1] * ncol + [DEGREE_QUOTIENT_POLYNOMIAL], n, pts)
    ze = [F.value(e[0, 0]) for e in evl[:ncol]]; zwe = [F.value(e[1, 0]) for e in evl[:ncol]]
    qz, qzw = F.values(evl[ncol][0]), F.values(evl[ncol][1])
    fr = khip.Sponge(khip.Sponge.FR, curve)
    fr.absorb(fq.digest())
    flat = [x for a, b in zip(ze, zwe) for x in (a, b)] + [x for a, b in zip(qz, qzw) for x in (a, b)]
    fr.absorb(F.limbs_many(flat))
    v = scalar_challenge(curve, F, fr.challenge())
    u = scalar_challenge(curve, F, fr.challenge())
    fr.free()
    # ---- one opening of all of them (blinder 1 per column, 1 per quotient chunk)
    a_dev = khip.DevBuf(NB); b_dev = khip.DevBuf(NB)
    khip.combine_polys_dev(fid, polys, [n] * ncol + [7 * n], [1] * ncol + [DEGREE_QUOTIENT_POLYNOMIAL], F.limbs(v), n, a_dev)
    khip.b_init_dev(fid, pts, F.limbs(u), n, b_dev)
    blinding, cip, ps = 0, 0, 1
    for e0, e1 in list(zip(ze, zwe)) + list(zip(qz, qzw)):
        blinding = (blinding + ps) % F.p
        cip = (cip + ps * ((e0 + u * e1) % F.p)) % F.p
        ps = ps * v % F.p
    bl = F.rand_many(rng, 2 * log2_n + 2)
    lr_xy, lr_inf, delta, dinf, z1, z2, sg, sg_inf = khip.ipa_open(srs, a_dev, b_dev, n, F.limbs(cip), F.limbs(blinding), fq_before, F.limbs_many(bl))
    fq_before.free(); fq.free()
    for b in (ev, cf, e8, t8, quot, rem, a_dev, b_dev):
        b.free()
    return {"commitments": [(cxy[i:i + 1], cinf[i:i + 1]) for i in range(ncol)], "zeta_evaluations": ze, "zeta_omega_evaluations": zwe,
            "quotient_commitment": (qxy, qci), "quotient_evaluations": (qz, qzw),
            "opening": {"lr": [(lr_xy[r], lr_inf[r]) for r in range(log2_n)], "delta": (delta, dinf), "z1": F.value(z1), "z2": F.value(z2), "sg": (sg, sg_inf)},
            "challenges": {"alpha": alpha, "zeta": zeta, "v": v, "u": u}}
