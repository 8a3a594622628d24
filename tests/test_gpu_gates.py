"""The gate library on the device: the token programs of proof_systems_amd/polish.py run by
kh_expr_evaluations_dev over device-resident columns equal the oracle's per-row machines (oracle/gates.py), on d1 and --
after iNTT + 8x extension -- on d8, where the combined constraint of a SATISFIED witness is divisible by the vanishing
polynomial (zero remainder) and that of a violated one is not: the quotient step of prover.rs:794-917 for each gate type."""
import random

import numpy as np
import pytest

from oracle import cref
from oracle import gates as G
from oracle import pasta as P
from proof_systems_amd import polish as OP

from test_gates import CURVE, F, gate_rows, kimchi_params, program, tables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def _limbs(vals):
    return cref.ints_to_limbs([F.to_mont(v) for v in vals])


def _ints(limbs):
    return [F.from_mont(v) for v in cref.limbs_to_ints(limbs)]


@pytest.mark.parametrize("name", list(OP.GATES))
def test_gate_on_device(khip, name):
    fid = khip.FP
    rnd = random.Random(1000 + sum(map(ord, name)))
    mds, _ = kimchi_params()
    endo = P.endos(CURVE)[0]
    w, co, ngate = tables(name, rnd)
    logn = 6; n = 1 << logn
    assert len(w) <= n - 3
    alpha = rnd.randrange(F.p)
    toks, consts = program(name, alpha)
    for variant in ("satisfied", "violated"):
        wt = [list(r) for r in w] + [[0] * 15 for _ in range(n - len(w))]
        ct = [list(r) for r in co] + [[0] * 15 for _ in range(n - len(co))]
        for r in range(n - 3, n):                                  # zero-knowledge rows: anything
            wt[r] = [rnd.randrange(F.p) for _ in range(15)]
        if variant == "violated":
            r0 = gate_rows(name, ngate)[len(gate_rows(name, ngate)) // 2]
            wt[r0][4] = (wt[r0][4] + 1) % F.p
        sel = [0] * n
        for r in gate_rows(name, ngate):
            sel[r] = 1
        cols = [[wt[r][c] for r in range(n)] for c in range(15)] + [[ct[r][c] for r in range(n)] for c in range(15)] + [sel]
        d1 = np.stack([_limbs(c) for c in cols])                   # (31, n, 4)
        bufs = [khip.DevBuf(n * 32).upload(d1[k]) for k in range(31)]
        out = khip.DevBuf(n * 32)
        khip.expr_evaluations_dev(fid, toks, bufs, [n] * 31, _limbs(consts), n, out, stride=1, next_shift=1)
        got = _ints(out.download((n, 4)))
        out2 = khip.DevBuf(n * 32)                                 # the compiled kernel of the same expression (csrc/gates.hip)
        khip.gate_evaluations_dev(fid, khip.gate_ids()[name], bufs, n, _limbs(consts), n, out2, stride=1, next_shift=1)
        assert _ints(out2.download((n, 4))) == got, (name, variant, "compiled kernel != token program")
        out2.free()
        want = [sel[r] * G.combined_row(F, name, wt[r], wt[(r + 1) % n], ct[r], alpha, mds=mds, endo=endo) % F.p for r in range(n)]
        assert got == want, (name, variant)
        if name == "ForeignFieldMul":                              # random rows (tests/test_gates.py): the program equals the row machine; nothing to divide
            assert any(want)
            for b in bufs + [out]:
                b.free()
            continue
        assert any(want) == (variant == "violated")
        # the quotient step: coefficient forms -> d8 -> constraint rows on d8 -> iNTT(8n) -> / Z_H
        coeffs = khip.ntt(fid, d1, logn, inverse=True)
        d8 = khip.lde(fid, coeffs, logn, 3)
        bufs8 = [khip.DevBuf(8 * n * 32).upload(d8[k]) for k in range(31)]
        t8 = khip.DevBuf(8 * n * 32)
        khip.expr_evaluations_dev(fid, toks, bufs8, [8 * n] * 31, _limbs(consts), 8 * n, t8, stride=1, next_shift=8)
        ev8 = t8.download((8 * n, 4))
        t8c = khip.DevBuf(8 * n * 32).upload(ev8)                  # compiled kernel, accumulating on top of the token program's rows: 2x
        khip.gate_evaluations_dev(fid, khip.gate_ids()[name], bufs8, 8 * n, _limbs(consts), 8 * n, t8c, stride=1, next_shift=8, accumulate=True)
        assert _ints(t8c.download((8 * n, 4))) == [2 * v % F.p for v in _ints(ev8)], (name, "compiled kernel on d8")
        t8c.free()
        assert _ints(ev8[::8]) == want                              # d1 is the stride-8 sub-grid of d8
        khip.ntt_dev(fid, t8, logn + 3, True, 1)
        q = khip.DevBuf(7 * n * 32); r = khip.DevBuf(n * 32)
        khip.divide_by_vanishing_poly_dev(fid, t8, 8 * n, logn, q, r)
        assert r.download((n, 4)).any() == (variant == "violated"), (name, variant)
        for b in bufs + bufs8 + [out, t8, q, r]:
            b.free()


@pytest.mark.parametrize("fid", [0, 1])
def test_generic_and_permutation_kernels(khip, fid):
    """The two arguments every circuit has, as compiled kernels ("Generic" on every second row of d8 into t4, "Permutation" on d8), equal
    their token lists run by the token machine -- which the oracle prover pins through the whole-proof parity tests."""
    Fd = P.Fp if fid == 0 else P.Fq
    rnd = random.Random(77 + fid)
    n = 1 << 9
    lim = lambda vals: cref.ints_to_limbs([Fd.to_mont(v) for v in vals])
    cols = [khip.DevBuf(n * 32).upload(lim([rnd.randrange(Fd.p) for _ in range(n)])) for _ in range(31)]
    gids = khip.gate_ids()
    assert khip.gate_num_constants(gids["Generic"]) == 2 and khip.gate_num_constants(gids["Permutation"]) == 10
    alpha = rnd.randrange(Fd.p)
    a = khip.DevBuf(n // 2 * 32); b = khip.DevBuf(n // 2 * 32)
    gen_cols = cols[:6] + cols[15:25] + [cols[30]]
    khip.expr_evaluations_dev(fid, OP.generic_gate_tokens(0, 6, 16, 0, 1), gen_cols, [n] * 17, lim([1, alpha]), n // 2, a, stride=2, next_shift=8)
    khip.gate_evaluations_dev(fid, gids["Generic"], cols, n, lim([1, alpha]), n // 2, b, stride=2, next_shift=8)
    assert (a.download((n // 2, 4)) == b.download((n // 2, 4))).all() and a.download((n // 2, 4)).any()
    pc = lim([rnd.randrange(Fd.p) for _ in range(10)])             # gamma, beta, alpha0, beta * shift_i
    perm_cols = cols[:7] + cols[15:22] + [cols[OP.PERM_Z_COL], cols[OP.PERM_X_COL], cols[OP.PERM_ZKPM_COL]]
    a2 = khip.DevBuf(n * 32); b2 = khip.DevBuf(n * 32)
    khip.expr_evaluations_dev(fid, OP.perm_quot_tokens(w0=0, s0=7, z=14, x=15, zkpm=16, gamma=0, beta=1, bshift0=3, alpha0=2), perm_cols, [n] * 17, pc, n, a2, stride=1, next_shift=8)
    khip.gate_evaluations_dev(fid, gids["Permutation"], cols, n, pc, n, b2, stride=1, next_shift=8)
    assert (a2.download((n, 4)) == b2.download((n, 4))).all() and a2.download((n, 4)).any()
    with pytest.raises(khip.KhError):                            # wrong constants count
        khip.gate_evaluations_dev(fid, gids["Permutation"], cols, n, pc[:9], n, b2)
    with pytest.raises(khip.KhError):                            # rows past the column
        khip.gate_evaluations_dev(fid, gids["Generic"], cols, n, lim([1, alpha]), n, b, stride=2)
    for x in cols + [a, b, a2, b2]:
        x.free()
