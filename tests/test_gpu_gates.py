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
        assert _ints(ev8[::8]) == want                              # d1 is the stride-8 sub-grid of d8
        khip.ntt_dev(fid, t8, logn + 3, True, 1)
        q = khip.DevBuf(7 * n * 32); r = khip.DevBuf(n * 32)
        khip.divide_by_vanishing_poly_dev(fid, t8, 8 * n, logn, q, r)
        assert r.download((n, 4)).any() == (variant == "violated"), (name, variant)
        for b in bufs + bufs8 + [out, t8, q, r]:
            b.free()
