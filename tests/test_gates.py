"""kimchi's gate library: the token programs of proof_systems_amd/polish.py against the oracle's per-row
machines (oracle/gates.py) -- CPU part, no GPU needed: the programs are run by the oracle's PolishToken machine.

For each of Poseidon, CompleteAdd, VarBaseMul, EndoMul, EndoMulScalar:
  * the reference's own witness generator (restated) zeroes every constraint on every gate row, and the gate computes what
    it claims (the Poseidon output equals the sponge permutation; the EC gates agree with the oracle's group law);
  * the compiled program evaluates to alpha-combined constraints identical to the straight formulas, row by row, on the
    satisfied table AND on a perturbed one (non-zero values must agree too);
  * perturbing any witness cell a constraint reads makes some constraint non-zero."""
import json
import os
import random

import pytest

from oracle import gates as G
from oracle import pasta as P
from proof_systems_amd import polish as OP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = P.Fp
CURVE = P.PALLAS                      # coordinates in Fp: the points a Vesta-side circuit computes on


def kimchi_params():
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon_kimchi_params.json")))["fp"]
    return [[int(x) for x in r] for r in d["mds"]], [[int(x) for x in r] for r in d["round_constants"]]


def tables(name, rnd):
    """(witness rows, coefficient rows, number of gate rows, extras) of a satisfied instance."""
    mds, rc = kimchi_params()
    endo = P.endos(CURVE)[0]
    if name == "Poseidon":
        st = [rnd.randrange(F.p) for _ in range(3)]
        w, co, out = G.poseidon_witness(F, st, mds, rc, 11)
        sp = __import__("oracle.poseidon", fromlist=["x"]).ArithmeticSponge(F); sp.state = list(st); sp.permute()
        assert sp.state == out                                   # 11 gate rows = the 55-round Kimchi permutation
        return w, co, 11
    if name == "CompleteAdd":
        pts = [CURVE.mul(CURVE.gen, rnd.randrange(1, 1 << 60)) for _ in range(6)]
        cases = [(pts[0], pts[1]), (pts[2], pts[2]), (pts[3], (pts[3][0], (-pts[3][1]) % F.p)), (pts[4], pts[5])]
        w = []
        for a, b in cases:
            row = G.complete_add_witness(F, a, b)
            want = CURVE.add(a, b)
            if want is None:
                assert row[6] == 1
            else:
                assert row[6] == 0 and (row[4], row[5]) == want
            w.append(row)
        return w + [[0] * 15], [[0] * 15] * 5, 4
    if name == "VarBaseMul":
        base = CURVE.mul(CURVE.gen, rnd.randrange(1, 1 << 60))
        bits = [rnd.randrange(2) for _ in range(20)]
        acc0 = CURVE.add(base, base)
        w, acc, n = G.varbasemul_witness(F, base, bits, acc0)
        k = 2                                                    # acc <- 2 acc + (2b - 1) base per bit, from acc0 = 2 base
        for b in bits:
            k = 2 * k + (2 * b - 1)
        assert acc == CURVE.mul(base, k) and n == int("".join(map(str, bits)), 2)
        return w + [[0] * 15], [[0] * 15] * (len(w) + 1), len(w)   # gate rows: every even row (the odd ones are read as `next`)
    if name == "EndoMul":
        base = CURVE.mul(CURVE.gen, rnd.randrange(1, 1 << 60))
        phi = (endo * base[0] % F.p, base[1])
        acc0 = CURVE.add(CURVE.add(base, phi), CURVE.add(base, phi))
        bits = [rnd.randrange(2) for _ in range(32)]
        w, acc, n = G.endomul_witness(F, endo, base, bits, acc0)
        want = acc0
        for i in range(0, 32, 2):                                # A <- 2A + Q, Q in {+-T, +-phi(T)} chosen by (b_x, b_sign)
            q = phi if bits[i] else base
            q = q if bits[i + 1] else (q[0], (-q[1]) % F.p)
            want = CURVE.add(CURVE.add(want, q), want)
        assert acc == want and n == int("".join(map(str, bits)), 2)
        return w, [[0] * 15] * len(w), len(w) - 1
    if name == "Xor16":
        a, b = rnd.randrange(1 << 64), rnd.randrange(1 << 64)
        w = G.xor_witness(F, a, b, 64)                           # 4 Xor16 rows + the zero row
        assert w[0][2] == a ^ b
        return w, [[0] * 15] * len(w), len(w) - 1
    if name == "RangeCheck0":                                      # two chained rows: the second one's compact form (coefficient 1) reads the third
        rows, co = [], []
        vals = [rnd.randrange(1 << 88) for _ in range(2)]
        for v in vals:
            rows.append([v] + [(v >> (76 - 12 * k)) & 4095 for k in range(6)] + [(v >> (14 - 2 * k)) & 3 for k in range(8)])
        rows.append([rnd.randrange(1 << 88), (vals[1] + (1 << 88) * 0) % F.p] + [0] * 13)
        rows[2][1] = (vals[1] + (1 << 88) * rows[2][0]) % F.p
        return rows, [[0] * 15, [1] + [0] * 14, [0] * 15], 2
    if name == "RangeCheck1":
        v = rnd.randrange(1 << 88)
        cur = [v, 0, (v >> 86) & 3] + [(v >> (74 - 12 * k)) & 4095 for k in range(4)] + [(v >> (36 - 2 * k)) & 3 for k in range(8)]
        nx = [(v >> (20 - 2 * k)) & 3 for k in range(3)] + [0] * 4 + [(v >> (14 - 2 * k)) & 3 for k in range(8)]
        return [cur, nx], [[0] * 15] * 2, 1
    if name == "Rot64":
        word, rot = rnd.randrange(1 << 64), 13
        excess = word >> (64 - rot); shifted = (word << rot) & ((1 << 64) - 1); bound = excess - (1 << rot) + (1 << 64)
        cur = [word, shifted + excess, excess] + [(bound >> (52 - 12 * k)) & 4095 for k in range(4)] + [(bound >> (14 - 2 * k)) & 3 for k in range(8)]
        return [cur, [shifted] + [0] * 14], [[1 << rot] + [0] * 14, [0] * 15], 1
    if name == "ForeignFieldAdd":                                  # a + b = r + overflow * f over three 88-bit limbs (secp256k1's base field as the foreign modulus)
        f = (1 << 256) - (1 << 32) - 977
        a, b = rnd.randrange(f), rnd.randrange(f)
        ovf = 1 if a + b >= f else 0
        r = a + b - ovf * f
        lim = lambda x: [x & ((1 << 88) - 1), (x >> 88) & ((1 << 88) - 1), x >> 176]
        al, bl, rl, fl = lim(a), lim(b), lim(r), lim(f)
        bot = al[0] + (al[1] << 88) + bl[0] + (bl[1] << 88) - ovf * (fl[0] + (fl[1] << 88)) - (rl[0] + (rl[1] << 88))
        carry = bot >> 176                                           # -1, 0 or 1
        assert bot == carry << 176
        cur = al + bl + [ovf, carry % F.p] + [0] * 7
        return [cur, rl + [0] * 12], [fl + [1] + [0] * 11, [0] * 15], 1
    if name == "ForeignFieldMul":                                  # no witness generator restated: the program is compared with the row machine on random rows only
        w = [[rnd.randrange(F.p) for _ in range(15)] for _ in range(3)]
        return w, [[rnd.randrange(F.p) for _ in range(15)] for _ in range(3)], 2
    scalar = rnd.randrange(1 << 128)
    w, _ = G.endomul_scalar_witness(F, scalar, endo, 128)
    return w + [[0] * 15], [[0] * 15] * (len(w) + 1), len(w)


def program(name, alpha):
    mds, _ = kimchi_params()
    return OP.gate_program(name, F.p, alpha, selector_col=30, mds=mds, endo=P.endos(CURVE)[0])


def gate_rows(name, ngate):
    return list(range(0, ngate, 2)) if name == "VarBaseMul" else list(range(ngate))


@pytest.mark.parametrize("name", list(OP.GATES))
def test_gate_program_matches_row_machine(name):
    rnd = random.Random(sum(map(ord, name)))
    mds, _ = kimchi_params()
    endo = P.endos(CURVE)[0]
    w, co, ngate = tables(name, rnd)
    nrows = len(w)
    alpha = rnd.randrange(F.p)
    toks, consts = program(name, alpha)
    sel = [0] * nrows
    for r in gate_rows(name, ngate):
        sel[r] = 1
    for variant in ("satisfied", "perturbed"):
        wt = [list(r) for r in w]
        if variant == "perturbed":
            for r in range(nrows):
                wt[r][rnd.randrange(15)] = rnd.randrange(F.p)
        cols = [[wt[r][c] for r in range(nrows)] for c in range(15)] + [[co[r][c] for r in range(nrows)] for c in range(15)] + [sel]
        got = P.polish_evaluate_rows(F, toks, cols, consts, nrows, 1, 1)
        nonzero = 0
        for r in range(nrows):
            want = sel[r] * G.combined_row(F, name, wt[r], wt[(r + 1) % nrows], co[r], alpha, mds=mds, endo=endo) % F.p
            assert got[r] == want, (name, variant, r)
            nonzero += want != 0
        if name != "ForeignFieldMul":                              # (its rows above are random: pinned on nine reference proofs instead, test_reference_fixtures.py)
            assert (nonzero == 0) == (variant == "satisfied"), (name, variant)


@pytest.mark.parametrize("name", list(OP.GATES))
def test_every_constrained_cell_matters(name):
    rnd = random.Random(7)
    mds, _ = kimchi_params()
    endo = P.endos(CURVE)[0]
    w, co, ngate = tables(name, rnd)
    row = gate_rows(name, ngate)[0]
    used = {"Poseidon": range(15), "CompleteAdd": range(11), "VarBaseMul": [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14], "EndoMul": [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14],
            "EndoMulScalar": range(14), "Xor16": range(15), "RangeCheck0": range(15), "RangeCheck1": [0] + list(range(2, 15)), "Rot64": [0, 1, 2] + list(range(3, 15)),
            "ForeignFieldAdd": range(8), "ForeignFieldMul": range(15)}[name]
    for c in used:
        wt = [list(r) for r in w]
        wt[row][c] = (wt[row][c] + 1 + rnd.randrange(5)) % F.p
        vals = [G.combined_row(F, name, wt[r], wt[(r + 1) % len(wt)], co[r], 3, mds=mds, endo=endo) for r in gate_rows(name, ngate)]
        assert any(vals), (name, c)


def test_generated_gate_kernels_are_current():
    """csrc/gates_gen.inc is the output of tools/gen_gate_kernels.py for the expressions of proof_systems_amd/polish.py as they are now."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_gate_kernels.py"), "--check"], cwd=root).returncode == 0


def test_compiled_expressions_keep_the_constants_layout():
    """The constants table a compiled kernel reads is laid out by the builder, independent of the values: per-proof values (challenges) take
    env.param slots that never merge with literals or with each other."""
    for p in (P.Fp.p, P.Fq.p):
        env = OP.Env(p)
        OP.permutation_expression(env, gamma=5, beta=5, alpha0=1, bshifts=[5] * 7)
        assert env.consts == [5, 5, 1] + [5] * 7
        env = OP.Env(p)
        OP.generic_expression(env, alpha=1)
        assert env.consts == [1, 1]


@pytest.mark.parametrize("fid", [0, 1])
def test_gate_constants_from_the_library_equal_the_builders(fid):
    """kh_gate_constants (host only: the recipe tools/gen_gate_kernels.py derived from the expression DAGs) returns, for every compiled gate, the
    constants table polish.gate_program builds for the same alpha / endo coefficient -- a C or Rust caller needs no copy of the builder."""
    import random
    import numpy as np
    from oracle import cref
    import proof_systems_amd.khip as khip
    Fd = P.Fp if fid == 0 else P.Fq
    rnd = random.Random(5 + fid)
    lim = lambda vals: cref.ints_to_limbs([Fd.to_mont(v % Fd.p) for v in vals])
    gids = khip.gate_ids()
    for name in OP.GATES:
        alpha, endo = rnd.randrange(Fd.p), rnd.randrange(Fd.p)
        _, consts = OP.gate_program(name, Fd.p, alpha, selector_col=30, mds=OP.POSEIDON_MDS[fid], endo=endo)
        got = khip.gate_constants(fid, gids[name], lim([alpha])[0], lim([endo])[0])
        assert (got == lim(consts)).all(), name
    params = [rnd.randrange(Fd.p) for _ in range(10)]
    assert (khip.gate_constants(fid, gids["Permutation"], params=lim(params)) == lim(params)).all()
    assert (khip.gate_constants(fid, gids["Generic"], params=lim(params[:2])) == lim(params[:2])).all()
    with pytest.raises(khip.KhError):
        khip.gate_constants(fid, gids["Permutation"], params=lim(params[:9]))
    with pytest.raises(khip.KhError):
        khip.gate_constants(fid, gids["Poseidon"])                       # needs alpha
