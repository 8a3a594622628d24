"""Row-parallel evaluation of compiled constraint expressions (csrc/expr.hip, SURVEY 8f rank 2 first slice) against the
oracle's per-row PolishToken machine (kimchi/src/circuits/expr.rs:856-937), and a miniature quotient pipeline:
witness/coefficient columns on d1 -> iNTT -> LDE to d8 -> generic-gate expression on d8 -> iNTT(8n) ->
divide_by_vanishing_poly: the remainder must be zero exactly when the gates are satisfied (prover.rs:794-910)."""
import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def _limbs(F, vals):
    return cref.ints_to_limbs([F.to_mont(v) for v in vals])


def _ints(F, limbs):
    return [F.from_mont(v) for v in cref.limbs_to_ints(limbs)]


def _rand(rnd, F, k):
    return [int.from_bytes(rnd.bytes(40), "little") % F.p for _ in range(k)]


@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_random_programs_match_oracle(khip, fid, F):
    rnd = np.random.default_rng(61 + fid)
    rows = 300                                             # three blocks, ragged tail
    cols = [_rand(rnd, F, 600) for _ in range(4)] + [_rand(rnd, F, 300)]
    consts = _rand(rnd, F, 5) + [0, 1]
    T = P
    cell = lambda c, nxt=0: (T.TOK_CELL, 2 * c + nxt)
    programs = [
        # (a * b' - c)^5 + const, Store/Load/Dup
        ([cell(0), cell(1, 1), (T.TOK_MUL, 0), cell(2), (T.TOK_SUB, 0), (T.TOK_STORE, 0), (T.TOK_POW, 5), (T.TOK_CONST, 2), (T.TOK_ADD, 0),
          (T.TOK_LOAD, 0), (T.TOK_DUP, 0), (T.TOK_MUL, 0), (T.TOK_ADD, 0)], 2, 8),
        # deep stack: ((((c0 + c1) * c2) - c3) * k0) with operands pushed first
        ([cell(3), cell(2), cell(1), cell(0), (T.TOK_ADD, 0), (T.TOK_MUL, 0), (T.TOK_SUB, 0), (T.TOK_CONST, 0), (T.TOK_MUL, 0)], 1, 8),
        # a lone constant, Pow(0) and Pow(1)
        ([(T.TOK_CONST, 4), (T.TOK_POW, 0), cell(4), (T.TOK_POW, 1), (T.TOK_ADD, 0)], 1, 3),
        ([(T.TOK_CONST, 5)], 1, 8),
        # 24 operands on the stack before the first operation: 96 KB of LDS (above the 64 KB default limit)
        ([cell(k % 4, k & 1) for k in range(24)] + [(T.TOK_MUL if k % 3 == 0 else T.TOK_ADD, 0) for k in range(23)], 1, 8),
    ]
    bufs = [khip.DevBuf(len(c) * 32).upload(_limbs(F, c)) for c in cols]
    out = khip.DevBuf(rows * 32)
    for toks, stride, shift in programs:
        khip.expr_evaluations_dev(fid, toks, bufs, [len(c) for c in cols], _limbs(F, consts), rows, out, stride=stride, next_shift=shift)
        want = P.polish_evaluate_rows(F, toks, cols, consts, rows, stride, shift)
        assert _ints(F, out.download((rows, 4))) == want
    # accumulate: out += value (t8 += eval, prover.rs:868-873)
    toks, stride, shift = programs[1]
    khip.expr_evaluations_dev(fid, toks, bufs, [len(c) for c in cols], _limbs(F, consts), rows, out, stride=stride, next_shift=shift)
    khip.expr_evaluations_dev(fid, programs[0][0], bufs, [len(c) for c in cols], _limbs(F, consts), rows, out, stride=2, next_shift=8, accumulate=True)
    a = P.polish_evaluate_rows(F, toks, cols, consts, rows, stride, shift)
    b = P.polish_evaluate_rows(F, programs[0][0], cols, consts, rows, 2, 8)
    assert _ints(F, out.download((rows, 4))) == [(x + y) % F.p for x, y in zip(a, b)]
    # malformed programs are rejected before launch
    too_deep = [cell(0)] * 45 + [(T.TOK_ADD, 0)] * 44      # 45 stack slots = 180 KB: more LDS than a CU has
    with pytest.raises(khip.KhError):
        khip.expr_evaluations_dev(fid, too_deep, bufs, [len(c) for c in cols], _limbs(F, consts), rows, out)
    for bad in ([(T.TOK_ADD, 0)], [cell(0), cell(1)], [cell(0), (T.TOK_LOAD, 0)], [cell(9)], [(T.TOK_CONST, 99)], [(42, 0)]):
        with pytest.raises(khip.KhError):
            khip.expr_evaluations_dev(fid, bad, bufs, [len(c) for c in cols], _limbs(F, consts), rows, out)
    for b in bufs + [out]:
        b.free()


def test_generic_gate_quotient_pipeline(khip):
    """Double generic gates on a domain of 64 rows: addition, multiplication and constant gates with a satisfying
    witness.  The combined constraint evaluated on d8 by the device vanishes on every 8th row, equals the oracle row
    for row, and f = iNTT_8n(evals) is divisible by x^n - 1; with one witness cell changed it is not."""
    F = P.Fp; fid = 0
    logn = 6; n = 1 << logn
    rnd = np.random.default_rng(71)
    w = [[0] * n for _ in range(6)]; c = [[0] * n for _ in range(10)]; sel = [0] * n
    for r in range(n - 3):                                  # last rows: no gate (selector 0), random witness
        sel[r] = 1
        for g in range(2):
            a, b = _rand(rnd, F, 2)
            kind = (r + g) % 3
            if kind == 0:                                  # a + b - o = 0
                co = [1, 1, F.p - 1, 0, 0]; o = (a + b) % F.p
            elif kind == 1:                                # a * b - o = 0
                co = [0, 0, F.p - 1, 1, 0]; o = a * b % F.p
            else:                                          # a - const = 0
                co = [1, 0, 0, 0, (-a) % F.p]; o = _rand(rnd, F, 1)[0]
            w[3 * g][r], w[3 * g + 1][r], w[3 * g + 2][r] = a, b, o
            for k in range(5):
                c[5 * g + k][r] = co[k]
    for r in range(n - 3, n):
        for k in range(6):
            w[k][r] = _rand(rnd, F, 1)[0]
    alpha = _rand(rnd, F, 1)[0]
    consts = [1, alpha]
    toks = P.generic_gate_tokens(w0=0, c0=6, sel=16, alpha0=0, alpha1=1)

    def pipeline(wit):
        d1 = np.stack([_limbs(F, col) for col in wit + c + [sel]])                 # 17 columns on d1
        coeffs = khip.ntt(fid, d1, logn, inverse=True)                             # interpolate (prover.rs:370-381)
        d8 = khip.lde(fid, coeffs, logn, 3)                                        # evaluate_over_domain(d8) (constraints.rs:490-495)
        bufs = [khip.DevBuf(8 * n * 32).upload(d8[k]) for k in range(17)]
        out = khip.DevBuf(8 * n * 32)
        khip.expr_evaluations_dev(fid, toks, bufs, [8 * n] * 17, _limbs(F, consts), 8 * n, out, stride=1, next_shift=8)
        ev = out.download((8 * n, 4))
        cols_int = [_ints(F, d8[k]) for k in range(17)]
        assert _ints(F, ev) == P.polish_evaluate_rows(F, toks, cols_int, consts, 8 * n, 1, 8)
        khip.ntt_dev(fid, out, logn + 3, True, 1)                                  # t8.interpolate() (prover.rs:907)
        q = khip.DevBuf(7 * n * 32); r = khip.DevBuf(n * 32)
        khip.divide_by_vanishing_poly_dev(fid, out, 8 * n, logn, q, r)             # prover.rs:903
        rem = r.download((n, 4))
        for b in bufs + [out, q, r]:
            b.free()
        return ev, rem

    ev, rem = pipeline(w)
    assert not ev[::8].any() and ev.any()                  # vanishes on d1, not identically zero on d8
    assert not rem.any()                                   # divisible by the vanishing polynomial
    w_bad = [list(col) for col in w]; w_bad[2][3] = (w_bad[2][3] + 1) % F.p      # row 3, gate 0 is an addition gate: its output matters
    ev, rem = pipeline(w_bad)
    assert ev[8 * 3].any() and rem.any()                   # "rest of division by vanishing polynomial" (prover.rs:904-908)
