"""Scans, batch inversion and division by a linear factor (csrc/poly.hip) against their definitions, and the permutation
argument of kimchi assembled from them on the device (perm_aggreg, permutation.rs:447-577; perm_quot, :216-331):
the accumulator z equals the oracle's literal loop, ends in 1, and both the permutation quotient and the two boundary
quotients divide exactly."""
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
def test_scans_and_batch_inversion(khip, fid, F):
    rnd = np.random.default_rng(81 + fid)
    for n in (1, 7, 2048, 2049, 5000, 70001):
        v = _rand(rnd, F, n)
        for op, f, ident in ((khip.SCAN_ADD, lambda a, b: (a + b) % F.p, 0), (khip.SCAN_MUL, lambda a, b: a * b % F.p, 1)):
            for rev in (False, True):
                d = khip.DevBuf(n * 32).upload(_limbs(F, v))
                khip.field_scan_dev(fid, op, d, n, reverse=rev)
                seq = v[::-1] if rev else v
                acc, want = ident, []
                for x in seq:
                    acc = f(acc, x); want.append(acc)
                if rev:
                    want = want[::-1]
                assert _ints(F, d.download((n, 4))) == want, (n, op, rev)
                d.free()
    n = 5000
    v = _rand(rnd, F, n)
    for i in (0, 17, 2047, 2048, 4999):
        v[i] = 0                                           # batch_inversion leaves zeros alone
    d = khip.DevBuf(n * 32).upload(_limbs(F, v))
    khip.batch_inversion_dev(fid, d, n)
    assert _ints(F, d.download((n, 4))) == [F.inv(x) if x else 0 for x in v]
    d.free()
    d = khip.DevBuf(32).upload(_limbs(F, [5]))
    khip.batch_inversion_dev(fid, d, 1)
    assert _ints(F, d.download((1, 4))) == [F.inv(5)]
    d.free()


@pytest.mark.parametrize("fid,F", [(0, P.Fp), (1, P.Fq)])
def test_divide_by_linear(khip, fid, F):
    rnd = np.random.default_rng(91 + fid)
    w = F.root_of_unity(10)
    for length in (1, 2, 1000, 4097):
        f = _rand(rnd, F, length)
        for a in (1, pow(w, 1021, F.p), 0, _rand(rnd, F, 1)[0]):
            fd = khip.DevBuf(length * 32).upload(_limbs(F, f)); qd = khip.DevBuf(max(length - 1, 1) * 32)
            rem = F.from_mont(P.from_limbs(khip.divide_by_linear_dev(fid, fd, length, _limbs(F, [a])[0], qd)))
            q = _ints(F, qd.download((length - 1, 4))) if length > 1 else []
            acc = 0
            for c in reversed(f):
                acc = (acc * a + c) % F.p
            assert rem == acc                              # f(a)
            back = [0] * length                            # q (x - a) + rem == f
            for i, v in enumerate(q):
                back[i + 1] = (back[i + 1] + v) % F.p
                back[i] = (back[i] - a * v) % F.p
            back[0] = (back[0] + rem) % F.p
            assert back == f
            # the asynchronous form: same quotient, the remainder stays on the device; kh_check_equal_dev flags it (or not) without a stall
            q2 = khip.DevBuf(max(length - 1, 1) * 32); rd = khip.DevBuf(32); flags = khip.DevBuf(4).upload(np.zeros(1, dtype=np.uint32))
            khip.divide_by_linear_async_dev(fid, fd, length, _limbs(F, [a])[0], q2, rd)
            khip.check_equal_dev(rd, 1, None, flags, 3)                        # bit 3: remainder != 0
            khip.check_equal_dev(rd, 1, _limbs(F, [rem])[0], flags, 4)        # bit 4: remainder != f(a)  (never)
            khip.check_equal_dev(fd, length, _limbs(F, [f[0]])[0], flags, 5)  # bit 5: some coefficient differs from the first
            khip.sync()
            assert _ints(F, rd.download((1, 4))) == [rem] and (length == 1 or q2.download((length - 1, 4)).tobytes() == qd.download((length - 1, 4)).tobytes())
            want = (8 if rem else 0) | (32 if any(c != f[0] for c in f) else 0)
            assert int(flags.download((1,), dtype=np.uint32)[0]) == want
            fd.free(); qd.free(); q2.free(); rd.free(); flags.free()


def test_permutation_argument_on_device(khip):
    F = P.Fp; fid = 0
    logn = 6; n = 1 << logn; zk = 3; PERMUTS = 7
    rnd = np.random.default_rng(101)
    omega = F.root_of_unity(logn)
    sid = [pow(omega, j, F.p) for j in range(n)]
    shifts = [pow(7, i, F.p) for i in range(PERMUTS)]       # distinct coset representatives (the test only needs them distinct)
    # a random permutation of the cells of the first n - zk rows, witness constant on its cycles
    cells = [(i, j) for i in range(PERMUTS) for j in range(n - zk)]
    perm = [cells[k] for k in rnd.permutation(len(cells))]
    pi = dict(zip(cells, perm))
    w = [[None] * n for _ in range(PERMUTS)]
    for cell in cells:
        if w[cell[0]][cell[1]] is None:
            v = _rand(rnd, F, 1)[0]
            cur = cell
            while w[cur[0]][cur[1]] is None:
                w[cur[0]][cur[1]] = v
                cur = pi[cur]
    sigma = [[0] * n for _ in range(PERMUTS)]
    for i in range(PERMUTS):
        for j in range(n):
            if j < n - zk:
                ti, tj = pi[(i, j)]
                sigma[i][j] = shifts[ti] * sid[tj] % F.p
            else:
                sigma[i][j] = shifts[i] * sid[j] % F.p
                w[i][j] = _rand(rnd, F, 1)[0]
    beta, gamma, alpha0 = _rand(rnd, F, 3)
    rands = _rand(rnd, F, 2)
    want_z = P.perm_aggreg(F, w, sigma, shifts, sid, beta, gamma, zk, rands)
    assert want_z[n - zk] == 1                              # "final value" check of the reference (permutation.rs:566-568)

    # ---- perm_aggreg on the device: numerators / denominators by the expression evaluator on d1, batch inversion, running product
    T = P
    cell = lambda c, nxt=0: (T.TOK_CELL, 2 * c + nxt)
    d1_cols = [khip.DevBuf(n * 32).upload(_limbs(F, col)) for col in w + sigma + [sid]]        # 0-6 w, 7-13 sigma, 14 sid
    consts = [gamma, beta] + [beta * s % F.p for s in shifts]                                   # 0 gamma, 1 beta, 2.. beta*shift_i
    num_t, den_t = [], []
    for i in range(PERMUTS):
        num_t += [cell(i), cell(14), (T.TOK_CONST, 2 + i), (T.TOK_MUL, 0), (T.TOK_ADD, 0), (T.TOK_CONST, 0), (T.TOK_ADD, 0)] + ([(T.TOK_MUL, 0)] if i else [])
        den_t += [cell(i), cell(7 + i), (T.TOK_CONST, 1), (T.TOK_MUL, 0), (T.TOK_ADD, 0), (T.TOK_CONST, 0), (T.TOK_ADD, 0)] + ([(T.TOK_MUL, 0)] if i else [])
    one = _limbs(F, [1])
    num = khip.DevBuf(n * 32).upload(one); den = khip.DevBuf(n * 32).upload(one)               # entry 0 = 1
    lens = [n] * 15
    khip.expr_evaluations_dev(fid, num_t, d1_cols, lens, _limbs(F, consts), n - 1, num, out_offset=1)
    khip.expr_evaluations_dev(fid, den_t, d1_cols, lens, _limbs(F, consts), n - 1, den, out_offset=1)
    khip.batch_inversion_dev(fid, den, n - 1, offset=1)                                        # permutation.rs:533
    ratio = khip.DevBuf(n * 32)
    khip.expr_evaluations_dev(fid, [cell(0), cell(1), (T.TOK_MUL, 0)], [num, den], [n, n], one, n, ratio)
    khip.field_scan_dev(fid, khip.SCAN_MUL, ratio, n - zk + 1)                                 # z[0 .. n - zk]
    z = _ints(F, ratio.download((n, 4)))
    z[n - zk + 1], z[n - zk + 2] = rands                   # the zero-knowledge rows (host RNG), permutation.rs:556-563
    assert z[: n - zk + 1] == want_z[: n - zk + 1] and z[n - zk] == 1
    z = want_z                                             # (with zk = 3 nothing follows the two random rows)
    assert n - zk + 2 == n - 1

    # ---- perm_quot on the device
    zkpm = [1] * (8 * n)                                    # permutation_vanishing_polynomial_l over d8: prod_{j >= n - zk} (x - w^j)
    om8 = F.root_of_unity(logn + 3)
    x8 = [pow(om8, i, F.p) for i in range(8 * n)]
    for i in range(8 * n):
        for j in range(n - zk, n):
            zkpm[i] = zkpm[i] * (x8[i] - sid[j]) % F.p
    d1 = np.stack([_limbs(F, col) for col in w + sigma + [z]])
    coeffs = khip.ntt(fid, d1, logn, inverse=True)
    d8 = khip.lde(fid, coeffs, logn, 3)
    cols8 = [khip.DevBuf(8 * n * 32).upload(d8[k]) for k in range(15)] + [khip.DevBuf(8 * n * 32).upload(_limbs(F, x8)),
                                                                         khip.DevBuf(8 * n * 32).upload(_limbs(F, zkpm))]
    consts = [gamma, beta, alpha0] + [beta * s % F.p for s in shifts]
    toks = P.perm_quot_tokens(w0=0, s0=7, z=14, x=15, zkpm=16, gamma=0, beta=1, bshift0=3, alpha0=2)
    out = khip.DevBuf(8 * n * 32)
    khip.expr_evaluations_dev(fid, toks, cols8, [8 * n] * 17, _limbs(F, consts), 8 * n, out, stride=1, next_shift=8)
    ev = out.download((8 * n, 4))
    cols_int = [_ints(F, d8[k]) for k in range(15)] + [x8, zkpm]
    assert _ints(F, ev) == P.polish_evaluate_rows(F, toks, cols_int, consts, 8 * n, 1, 8)
    assert not ev[::8].any() and ev.any()
    khip.ntt_dev(fid, out, logn + 3, True, 1)
    q = khip.DevBuf(7 * n * 32); r = khip.DevBuf(n * 32)
    khip.divide_by_vanishing_poly_dev(fid, out, 8 * n, logn, q, r)
    assert not r.download((n, 4)).any()                    # the permutation constraint is divisible by Z_H
    # boundary quotients: (z - 1) / (x - 1) and (z - 1) / (x - sid[n - zk]) (permutation.rs:291-327)
    zc = _ints(F, coeffs[14]); zc[0] = (zc[0] - 1) % F.p
    zm1 = khip.DevBuf(n * 32).upload(_limbs(F, zc)); bq = khip.DevBuf(n * 32)
    for a in (1, sid[n - zk]):
        rem = khip.divide_by_linear_dev(fid, zm1, n, _limbs(F, [a])[0], bq)
        assert not rem.any()                               # "first / second division rest"
    # an inconsistent witness breaks the accumulator: z[n - zk] != 1
    w_bad = [list(col) for col in w]; w_bad[3][10] = (w_bad[3][10] + 1) % F.p
    assert P.perm_aggreg(F, w_bad, sigma, shifts, sid, beta, gamma, zk, rands)[n - zk] != 1
    for b in d1_cols + cols8 + [num, den, ratio, out, q, r, zm1, bq]:
        b.free()
