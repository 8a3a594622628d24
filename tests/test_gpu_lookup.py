"""The lookup argument on the device (proof_systems_amd/lookup.py + the token program of polish.lookup_program) against the
oracle's restatement (oracle/lookup.py): index columns, the combined table, the sorted columns, the aggregation (expression
rows + batch inversion + running product), the constraint rows on d1, and -- after iNTT + 8x extension -- the quotient step:
the combined lookup constraints of a satisfied witness are divisible by the vanishing polynomial, those of a tampered
aggregation are not (prover.rs:874-917 for the lookup argument)."""
import random

import numpy as np
import pytest

from oracle import cref
from oracle import lookup as L
from oracle import pasta as P

from test_lookup import F, LOGN, N, ZK, circuit, p, user_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import proof_systems_amd.khip as khip
    from proof_systems_amd import lookup as LK
    from proof_systems_amd import polish as OP
    khip.init(0)
    return khip, LK, OP


def _limbs(vals):
    return cref.ints_to_limbs([F.to_mont(v) for v in vals])


def _ints(limbs):
    return [F.from_mont(v) for v in cref.limbs_to_ints(limbs)]


def test_lookup_argument_on_device(env):
    khip, LK, OP = env
    fid = khip.FP
    rnd = random.Random(11)
    gates, wit = circuit(rnd)
    cs = L.LookupCS(p, gates, [user_table()], N, ZK)
    ix = LK.LookupIndex(fid, gates, [user_table()], LOGN, ZK)
    # ---- index
    assert ix.patterns == cs.info.patterns and ix.max_per_row == cs.info.max_per_row and ix.max_joint_size == cs.info.max_joint_size
    assert ix.table_cols == cs.table_cols and ix.table_ids == cs.table_ids and ix.selectors == cs.selectors
    jc, beta, gamma, alpha = (rnd.randrange(p) for _ in range(4))
    d_table = ix.joint_table_dev(jc)
    table = cs.joint_table(jc)
    assert _ints(d_table.download((N, 4))) == table
    # ---- sorted (host in the reference and here), zero-knowledge rows
    want_sorted = L.sorted_columns(cs, gates, wit, jc)
    got_sorted = LK.sorted_columns(ix, wit, table, jc)
    assert got_sorted == want_sorted
    sorted_cols = [L.zk_patch(c, N, ZK, [rnd.randrange(p) for _ in range(ZK)]) for c in want_sorted]
    L.verify(cs, gates, wit, jc, sorted_cols)
    d_wit = [khip.DevBuf(N * 32).upload(_limbs(c)) for c in wit]
    d_sorted = [khip.DevBuf(N * 32).upload(_limbs(c)) for c in sorted_cols]
    # ---- aggregation on the device
    nrng = np.random.default_rng(3)
    d_agg = LK.aggregation_dev(ix, d_wit, d_sorted, d_table, jc, beta, gamma, nrng)
    agg = _ints(d_agg.download((N, 4)))
    want_agg = L.aggregation(cs, gates, wit, jc, beta, gamma, sorted_cols, agg[N - ZK:])
    assert agg == want_agg and agg[N - ZK - 1] == 1
    # ---- constraint rows on d1: token program vs the oracle's row machine
    cols = LK.column_layout(ix)
    _, tic = ix.constraint_combiners(jc)
    toks, consts = OP.lookup_program(p, ix.patterns, cols, jc, tic, beta, gamma, alpha)
    atoms1 = LK.atom_columns(ix, 0)
    omega = F.root_of_unity(LOGN)

    def run(d_w, d_s, d_a, d_t, atoms, rows, shift):
        bufs = list(d_w) + list(d_s) + [d_a, d_t] + [sel[q] for q in ix.patterns] + list(atoms)
        assert len(bufs) == cols["count"]
        out = khip.DevBuf(rows * 32)
        khip.expr_evaluations_dev(fid, toks, bufs, [rows] * len(bufs), _limbs(consts), rows, out, stride=1, next_shift=shift)
        return out
    sel = ix.d_selectors
    out = run(d_wit, d_sorted, d_agg, d_table, atoms1, N, 1)
    got = _ints(out.download((N, 4)))
    at1 = [_ints(a.download((N, 4))) for a in atoms1]

    def oracle_rows(w, s, a):
        res = []
        for r in range(N):
            colsv = {"w": w, "sorted": s, "aggreg": [a], "table": [table]}

            def cell(kind, idx, row, r=r):
                if kind == "selector":
                    return cs.selectors[idx][(r + row) % N]
                return colsv[kind][idx][(r + row) % N]
            x = pow(omega, r, p)
            atoms = {"vanish": at1[0][r], "l0": at1[1][r], "lfinal": at1[2][r]}
            assert atoms["vanish"] == L.vanishes_on_last_n_rows(p, omega, N, ZK + 1, x)
            vals = L.constraint_values(cs, jc, beta, gamma, cell, atoms)
            res.append(sum(pow(alpha, i, p) * v for i, v in enumerate(vals)) % p)
        return res
    assert got == oracle_rows(wit, sorted_cols, agg) and not any(got)
    bad = list(agg); bad[77] = (bad[77] + 5) % p
    d_bad = khip.DevBuf(N * 32).upload(_limbs(bad))
    out_bad = run(d_wit, d_sorted, d_bad, d_table, atoms1, N, 1)
    got_bad = _ints(out_bad.download((N, 4)))
    assert got_bad == oracle_rows(wit, sorted_cols, bad) and [r for r in range(N) if got_bad[r]] == [76, 77]
    # ---- the quotient step on d8
    atoms8 = LK.atom_columns(ix, 3)
    xs = [pow(F.root_of_unity(LOGN + 3), k, p) for k in (1, 9, 8 * N - 3)]
    a8 = [_ints(a.download((8 * N, 4))) for a in atoms8]
    for k, x in zip((1, 9, 8 * N - 3), xs):
        assert a8[0][k] == L.vanishes_on_last_n_rows(p, omega, N, ZK + 1, x)
        assert a8[1][k] == L.unnormalized_lagrange_basis(p, omega, N, 0, x)
        assert a8[2][k] == L.unnormalized_lagrange_basis(p, omega, N, -(ZK + 1), x)

    def to_d8(buf):
        c = khip.ntt(fid, buf.download((N, 4))[None], LOGN, inverse=True)
        return khip.DevBuf(8 * N * 32).upload(khip.lde(fid, c, LOGN, 3)[0])
    w8 = [to_d8(b) for b in d_wit]; s8 = [to_d8(b) for b in d_sorted]; t8 = to_d8(d_table)
    sel8 = {q: to_d8(sel[q]) for q in ix.patterns}
    for variant, d_a in (("satisfied", d_agg), ("violated", d_bad)):
        a8buf = to_d8(d_a)
        bufs = w8 + s8 + [a8buf, t8] + [sel8[q] for q in ix.patterns] + atoms8
        q8 = khip.DevBuf(8 * N * 32)
        khip.expr_evaluations_dev(fid, toks, bufs, [8 * N] * len(bufs), _limbs(consts), 8 * N, q8, stride=1, next_shift=8)
        ev = q8.download((8 * N, 4))
        assert _ints(ev[::8]) == (got if variant == "satisfied" else got_bad)
        khip.ntt_dev(fid, q8, LOGN + 3, True, 1)
        quo = khip.DevBuf(7 * N * 32); rem = khip.DevBuf(N * 32)
        khip.divide_by_vanishing_poly_dev(fid, q8, 8 * N, LOGN, quo, rem)
        assert rem.download((N, 4)).any() == (variant == "violated"), variant


@pytest.mark.parametrize("zk", [3, 5])
def test_atom_columns_on_the_device_equal_the_host_restatement(zk):
    """VanishesOnZeroKnowledgeAndPreviousRows, UnnormalizedLagrangeBasis(0), UnnormalizedLagrangeBasis(-zk_rows - 1) on d8 (expr.rs:883-893): the device
    version (products / one batched inversion per atom / the removable singularity patched) against the host loop over every point."""
    import proof_systems_amd.khip as khip
    from proof_systems_amd import lookup as LK, prover
    khip.init(0)
    logn = 7; n = 1 << logn
    F = prover.Fld(khip.FP)
    tables = [{"id": 0, "data": [list(range(8)), [0] + list(range(11, 18))]}]
    gates = ["Lookup"] * 20 + ["Zero"] * (n - zk - 20)
    LI = LK.LookupIndex(khip.FP, gates, tables, logn, zk_rows=zk)
    root = LK.khip_root(F, logn + 3)
    xs = [1] * (8 * n)
    for k in range(1, 8 * n):
        xs[k] = xs[k - 1] * root % F.p
    x8 = khip.DevBuf(8 * n * 32).upload(F.limbs_many(xs))
    host = LK.atom_columns(LI, 3)
    dev = LK.atom_columns_dev(LI, x8, 3)
    for a, b, name in zip(host, dev, ("vanish", "l0", "lfinal")):
        assert (a.download((8 * n, 4)) == b.download((8 * n, 4))).all(), name
    for b in host + dev + [x8]:
        b.free()
    LI.free()
