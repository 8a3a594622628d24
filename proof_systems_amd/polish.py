"""Reverse-Polish token streams (include/kimchi_hip.h KH_TOK_*) of the constraints this repository restates from the
reference, for callers of kh_expr_evaluations_dev that have no Rust `Expr::to_polish()` at hand (bench.py, tools/):
the double generic gate (kimchi/src/circuits/polynomials/generic.rs:83-120, argument.rs:201-214) and the `perm` part of
perm_quot (polynomials/permutation.rs:237-283)."""
TOK_CONST, TOK_CELL, TOK_DUP, TOK_POW, TOK_ADD, TOK_MUL, TOK_SUB, TOK_STORE, TOK_LOAD = range(9)


def cell(col: int, nxt: int = 0):
    return (TOK_CELL, 2 * col + nxt)


def generic_gate_tokens(w0: int, c0: int, sel: int, alpha0: int, alpha1: int):
    """index(Generic) * (alpha^a0 * constraint1 + alpha^a1 * constraint2); witness columns w0..w0+5, coefficient columns
    c0..c0+9, selector column sel, alpha powers at constants alpha0 / alpha1."""
    t = [cell(sel)]
    for g, alpha in ((0, alpha0), (1, alpha1)):
        w, c = w0 + 3 * g, c0 + 5 * g
        t += [(TOK_CONST, alpha)]
        t += [cell(c), cell(w), (TOK_MUL, 0)]
        t += [cell(c + 1), cell(w + 1), (TOK_MUL, 0), (TOK_ADD, 0)]
        t += [cell(c + 2), cell(w + 2), (TOK_MUL, 0), (TOK_ADD, 0)]
        t += [cell(c + 3), cell(w), (TOK_MUL, 0), cell(w + 1), (TOK_MUL, 0), (TOK_ADD, 0)]
        t += [cell(c + 4), (TOK_ADD, 0)]
        t += [(TOK_MUL, 0)]
        if g == 1:
            t += [(TOK_ADD, 0)]
    t += [(TOK_MUL, 0)]
    return t


def perm_quot_tokens(w0: int, s0: int, z: int, x: int, zkpm: int, gamma: int, beta: int, bshift0: int, alpha0: int, permuts: int = 7):
    """alpha0 * zkpm(x) * (z(x) prod_i (w_i + gamma + x beta shift_i) - z(x w) prod_i (w_i + gamma + sigma_i beta))."""
    t = []
    for i in range(permuts):
        t += [cell(w0 + i), (TOK_CONST, gamma), (TOK_ADD, 0), cell(x), (TOK_CONST, bshift0 + i), (TOK_MUL, 0), (TOK_ADD, 0)]
        if i:
            t += [(TOK_MUL, 0)]
    t += [cell(z), (TOK_MUL, 0)]
    for i in range(permuts):
        t += [cell(w0 + i), (TOK_CONST, gamma), cell(s0 + i), (TOK_CONST, beta), (TOK_MUL, 0), (TOK_ADD, 0), (TOK_ADD, 0)]
        if i:
            t += [(TOK_MUL, 0)]
    t += [cell(z, 1), (TOK_MUL, 0), (TOK_SUB, 0), (TOK_CONST, alpha0), (TOK_MUL, 0), cell(zkpm), (TOK_MUL, 0)]
    return t


def perm_aggreg_tokens(permuts: int = 7, w0: int = 0, s0: int = 7, sid: int = 14, gamma: int = 0, beta: int = 1, bshift0: int = 2):
    """(numerator, denominator) rows of perm_aggreg (permutation.rs:510-551): prod_i (w_i + sid beta shift_i + gamma) and
    prod_i (w_i + sigma_i beta + gamma)."""
    num, den = [], []
    for i in range(permuts):
        num += [cell(w0 + i), cell(sid), (TOK_CONST, bshift0 + i), (TOK_MUL, 0), (TOK_ADD, 0), (TOK_CONST, gamma), (TOK_ADD, 0)] + ([(TOK_MUL, 0)] if i else [])
        den += [cell(w0 + i), cell(s0 + i), (TOK_CONST, beta), (TOK_MUL, 0), (TOK_ADD, 0), (TOK_CONST, gamma), (TOK_ADD, 0)] + ([(TOK_MUL, 0)] if i else [])
    return num, den


# ======================================================================================================================
# A small expression builder (the role of kimchi's Expr + Expr::to_polish for callers without a Rust toolchain) and the
# gate library lowered with it: Poseidon, CompleteAdd, VarBaseMul, EndoMul, EndoMulScalar.  Each gate function restates
# `Argument::constraint_checks` of the reference (file:line in its docstring) on an environment that hands out cells;
# `combined_constraints` is `Argument::combined_constraints` (argument.rs:201-214): index(gate) * sum_i alpha^(a0 + i) c_i.
# ======================================================================================================================
class Node:
    """A node of an expression DAG.  `cached` nodes are emitted once (Store) and re-read afterwards (Load), like
    `Cache::cache` / PolishToken::Store / Load in the reference (expr.rs:815-836)."""
    __slots__ = ("op", "a", "b", "arg", "cached", "slot")

    def __init__(self, op, a=None, b=None, arg=0):
        self.op, self.a, self.b, self.arg, self.cached, self.slot = op, a, b, arg, False, None

    def _bin(self, op, o):
        return Node(op, self, o if isinstance(o, Node) else _lit(o))

    def __add__(self, o): return self._bin(TOK_ADD, o)
    def __sub__(self, o): return self._bin(TOK_SUB, o)
    def __mul__(self, o): return self._bin(TOK_MUL, o)
    def __radd__(self, o): return _lit(o)._bin(TOK_ADD, self)
    def __rsub__(self, o): return _lit(o)._bin(TOK_SUB, self)
    def __rmul__(self, o): return _lit(o)._bin(TOK_MUL, self)
    def __pow__(self, e): return Node(TOK_POW, self, None, int(e))
    def double(self): return Node("dbl", self)          # x Dup Add  (Expr::Double)
    def square(self): return Node("sqr", self)          # x Dup Mul  (Expr::Square)

    def cache(self):
        self.cached = True
        return self


class _Lit(Node):
    pass


def _lit(v):
    n = _Lit("lit"); n.arg = int(v)
    return n


class Env:
    """Hands out the cells and constants of one gate evaluation.  Column numbering of the token program: witness columns
    w0 .. w0+14, coefficient columns c0 .. c0+14, then the caller's extra columns (selectors).  Constants (literals, MDS
    entries, the endo coefficient, alpha powers) are collected in `self.consts` as integers mod p; `table(F_limbs)` turns
    them into the limb table kh_expr_evaluations_dev takes."""

    def __init__(self, p: int, w0: int = 0, c0: int = 15, mds=None, endo: int = 0):
        self.p, self.w0, self.c0, self._mds, self._endo = p, w0, c0, mds, endo
        self.consts, self._index, self.param_slots = [], {}, set()

    def const(self, v: int) -> Node:
        v %= self.p
        if v not in self._index:
            self._index[v] = len(self.consts); self.consts.append(v)
        return Node(TOK_CONST, arg=self._index[v])

    def param(self, v: int = 0) -> Node:
        """a constant slot of its own (a per-proof value: a challenge, a power of alpha): never merged with an equal literal, so that the
        LAYOUT of the constants table does not depend on the values"""
        self.consts.append(v % self.p)
        self.param_slots.add(len(self.consts) - 1)
        return Node(TOK_CONST, arg=len(self.consts) - 1)

    def witness_curr(self, i): return Node(TOK_CELL, arg=2 * (self.w0 + i))
    def witness_next(self, i): return Node(TOK_CELL, arg=2 * (self.w0 + i) + 1)
    def coeff(self, i): return Node(TOK_CELL, arg=2 * (self.c0 + i))
    def column(self, col): return Node(TOK_CELL, arg=2 * col)
    def column_next(self, col): return Node(TOK_CELL, arg=2 * col + 1)
    def mds(self, r, c): return self.const(self._mds[r][c])
    def endo_coefficient(self): return self.const(self._endo)
    def one(self): return self.const(1)


def compile_tokens(env: Env, expr: Node):
    """Postfix token list of `expr`; literals met on the way are entered into env.consts."""
    out, nslots = [], [0]

    def emit(n):
        if isinstance(n, _Lit):
            emit(env.const(n.arg)); return
        if n.cached and n.slot is not None:
            out.append((TOK_LOAD, n.slot)); return
        if n.op in (TOK_CONST, TOK_CELL):
            out.append((n.op, n.arg))
        elif n.op == TOK_POW:
            emit(n.a); out.append((TOK_POW, n.arg))
        elif n.op == "dbl":
            emit(n.a); out.append((TOK_DUP, 0)); out.append((TOK_ADD, 0))
        elif n.op == "sqr":
            emit(n.a); out.append((TOK_DUP, 0)); out.append((TOK_MUL, 0))
        else:
            emit(n.a); emit(n.b); out.append((n.op, 0))
        if n.cached:
            n.slot = nslots[0]; nslots[0] += 1
            out.append((TOK_STORE, 0))
    emit(expr)
    return out


def combined_constraints(env: Env, selector_col, constraints, alpha: int, alpha0: int = 0):
    """index(gate) * sum_i alpha^(alpha0 + i) * constraint_i (Expr::combine_constraints + the selector, argument.rs:201-214)."""
    acc = None
    for i, c in enumerate(constraints):
        term = env.const(pow(alpha, alpha0 + i, env.p)) * c
        acc = term if acc is None else acc + term
    return env.column(selector_col) * acc if selector_col is not None else acc


# ---------------------------------------------------------------------------------------------------------------- gates
ROUND_TO_COLS = [0, 2, 3, 4, 1]           # STATE_ORDER (poseidon.rs:65-73): round r lives in columns 3 * slot .. 3 * slot + 2


def poseidon_constraints(env: Env):
    """Poseidon::constraint_checks (kimchi/src/circuits/polynomials/poseidon.rs:351-436): 5 rounds per row, 15 constraints:
    state_(r+1)[j] - (rc[3 r + j] + sum_k mds[j][k] * state_r[k]^7); the fifth round's target is the next row's columns 0..2."""
    res = []
    for r in range(5):
        src = [env.witness_curr(3 * ROUND_TO_COLS[r] + k) for k in range(3)]
        sboxed = [(x ** 7).cache() for x in src]
        for j in range(3):
            tgt = env.witness_next(j) if r == 4 else env.witness_curr(3 * ROUND_TO_COLS[r + 1] + j)
            acc = env.coeff(3 * r + j)
            for k in range(3):
                acc = acc + env.mds(j, k) * sboxed[k]
            res.append(tgt - acc)
    return res


def complete_add_constraints(env: Env):
    """CompleteAdd::constraint_checks (complete_add.rs:103-226): 7 constraints on (x1, y1, x2, y2, x3, y3, inf, same_x, s, inf_z, x21_inv)."""
    x1, y1, x2, y2, x3, y3 = (env.witness_curr(i) for i in range(6))
    inf, same_x, s, inf_z, x21_inv = (env.witness_curr(i) for i in range(6, 11))
    x21 = (x2 - x1).cache(); y21 = (y2 - y1).cache()
    res = [x21_inv * x21 - (env.one() - same_x), same_x * x21]                       # zero_check(x21, x21_inv, same_x)
    x1_sq = (x1 * x1).cache()
    dbl_case = s.double() * y1 - x1_sq.double() - x1_sq
    add_case = x21 * s - y21
    res.append(same_x * dbl_case + (env.one() - same_x) * add_case)
    res.append(x1 + x2 + x3 - s * s)
    res.append(s * (x1 - x3) - y1 - y3)
    res.append(y21 * (same_x - inf))
    res.append(y21 * inf_z - inf)
    return res


def _single_bit(env: Env, b, base, s1, inp, out):
    """varbasemul.rs:226-277: one double-and-add step, 4 constraints."""
    b_sign = b.double() - env.one()
    s1_sq = (s1 * s1).cache()
    rx = s1_sq - inp[0] - base[0]
    t = (inp[0] - rx).cache()
    u = (inp[1].double() - t * s1).cache()
    return [b * b - b,
            (inp[0] - base[0]) * s1 - (inp[1] - b_sign * base[1]),
            u * u - (t * t) * (out[0] - base[0] + s1_sq),
            (out[1] + inp[1]) * t - (inp[0] - out[0]) * u]


def varbasemul_constraints(env: Env):
    """VarbaseMul::constraint_checks (varbasemul.rs:419-455; layout :312-331): 21 constraints over two rows.
    row i:   xT yT x0 y0 n n' . x1 y1 x2 y2 x3 y3 x4 y4      row i+1: x5 y5 b0 b1 b2 b3 b4 s0 s1 s2 s3 s4"""
    wc, wn = env.witness_curr, env.witness_next
    accs = [(wc(2), wc(3)), (wc(7), wc(8)), (wc(9), wc(10)), (wc(11), wc(12)), (wc(13), wc(14)), (wn(0), wn(1))]
    bits = [wn(2 + i) for i in range(5)]
    ss = [wn(7 + i) for i in range(5)]
    base = (wc(0), wc(1))
    acc = wc(4)
    for b in bits:
        acc = b + acc.double()
    res = [wc(5) - acc]
    for i in range(5):
        res += _single_bit(env, bits[i], base, ss[i], accs[i], accs[i + 1])
    return res


def endomul_constraints(env: Env):
    """EndosclMul::constraint_checks (endosclmul.rs:475-558): 12 constraints, 4 scalar bits per row.
    row i: xT yT inv . xP yP n xR yR s1 s3 b1 b2 b3 b4;  next row: xS = w4', yS = w5', n' = w6'."""
    wc, wn = env.witness_curr, env.witness_next
    b1, b2, b3, b4 = wc(11), wc(12), wc(13), wc(14)
    xt, yt, inv = wc(0), wc(1), wc(2)
    xs, ys = wn(4), wn(5)
    xp, yp, xr, yr, s1, s3 = wc(4), wc(5), wc(7), wc(8), wc(9), wc(10)
    endo_m1 = env.endo_coefficient() - env.one()
    xq1 = ((env.one() + b1 * endo_m1) * xt).cache()
    xq2 = ((env.one() + b3 * endo_m1) * xt).cache()
    yq1 = (b2.double() - env.one()) * yt
    yq2 = (b4.double() - env.one()) * yt
    s1_sq = (s1 * s1).cache(); s3_sq = (s3 * s3).cache()
    n, n_next = wc(6), wn(6)
    n_constraint = (((n.double() + b1).double() + b2).double() + b3).double() + b4 - n_next
    xp_xr = (xp - xr).cache(); xr_xs = (xr - xs).cache()
    ys_yr = (ys + yr).cache(); yr_yp = (yr + yp).cache()
    return [b1 * b1 - b1, b2 * b2 - b2, b3 * b3 - b3, b4 * b4 - b4,
            (xq1 - xp) * s1 - (yq1 - yp),
            ((xp.double() - s1_sq) + xq1) * ((xp_xr * s1) + yr_yp) - (yp.double() * xp_xr),
            yr_yp.square() - (xp_xr.square() * ((s1_sq - xq1) + xr)),
            (xq2 - xr) * s3 - (yq2 - yr),
            ((xr.double() - s3_sq) + xq2) * ((xr_xs * s3) + ys_yr) - (yr.double() * xr_xs),
            ys_yr.square() - (xr_xs.square() * ((s3_sq - xq2) + xs)),
            n_constraint,
            xp_xr * xr_xs * inv - env.one()]


def _polynomial(env: Env, coeffs, x):
    """Horner: sum_i coeffs[i] x^i (endomul_scalar.rs: `polynomial`)."""
    acc = None
    for c in reversed(coeffs):
        acc = env.const(c) if acc is None else acc * x + env.const(c)
    return acc


def endomul_scalar_constraints(env: Env):
    """EndomulScalar::constraint_checks (endomul_scalar.rs:174-222): 11 constraints, 8 crumbs per row
    (n0, n8, a0, b0, a8, b8, x0..x7 in columns 0..13)."""
    p = env.p
    inv = lambda v: pow(v, -1, p)
    wc = env.witness_curr
    n0, n8, a0, b0, a8, b8 = (wc(i) for i in range(6))
    xs = [wc(6 + i) for i in range(8)]
    c_coeffs = [0, 11 * inv(6) % p, (-5 * inv(2)) % p, 2 * inv(3) % p]
    crumb_over_x = [(-6) % p, 11, (-6) % p, 1]
    d_minus_c = [(-1) % p, 3, (-1) % p]
    c_funcs = [_polynomial(env, c_coeffs, x).cache() for x in xs]
    d_funcs = [c_funcs[i] + _polynomial(env, d_minus_c, xs[i]) for i in range(8)]
    n8_exp = n0
    for x in xs:
        n8_exp = n8_exp.double().double() + x
    a8_exp = a0
    for c in c_funcs:
        a8_exp = a8_exp.double() + c
    b8_exp = b0
    for d in d_funcs:
        b8_exp = b8_exp.double() + d
    return [n8_exp - n8, a8_exp - a8, b8_exp - b8] + [_polynomial(env, crumb_over_x, x) * x for x in xs]


def xor16_constraints(env: Env):
    """Xor16::constraint_checks (xor.rs:152-174): in1, in2, out (columns 0, 1, 2) each equal their four 4-bit nybbles
    (columns 3 + 4i .. 6 + 4i) + 2^16 * the next row's value; the XOR itself is the row's four lookups."""
    wc, wn = env.witness_curr, env.witness_next
    out = []
    for i in range(3):
        acc = wc(3 + 4 * i) + wc(4 + 4 * i) * env.const(1 << 4) + wc(5 + 4 * i) * env.const(1 << 8) + wc(6 + 4 * i) * env.const(1 << 12)
        out.append(acc + env.const(1 << 16) * wn(i) - wc(i))
    return out


LIMB_BITS = 88                               # KimchiForeignElement: three 88-bit limbs


def _crumb(env: Env, x):
    """constraints::crumb (expr.rs:3406-3412): x (x - 1)(x - 2)(x - 3)"""
    return x * (x - 1) * (x - 2) * (x - 3)


def _weighted_sum(env: Env, cells_and_bits):
    """sum of cells times increasing powers of two: [(cell, bits of the cell), ...] from the least significant one"""
    acc, shift = None, 0
    for c, bits in cells_and_bits:
        term = c if shift == 0 else env.const(1 << shift) * c
        acc = term if acc is None else acc + term
        shift += bits
    return acc


def range_check0_constraints(env: Env):
    """RangeCheck0::constraint_checks (range_check/circuitgates.rs:117-163): 8 crumbs, the 88-bit decomposition of column 0 (six 12-bit
    limbs in columns 1..6 -- range-checked by the row's lookups -- and the crumbs), and coeff 0 * (next[1] - (curr[0] + 2^88 next[0]))."""
    wc, wn = env.witness_curr, env.witness_next
    out = [_crumb(env, wc(i)) for i in range(7, 15)]
    limbs = [(wc(i), 2) for i in range(14, 6, -1)] + [(wc(i), 12) for i in range(6, 0, -1)]
    out.append(_weighted_sum(env, limbs) - wc(0))
    out.append(env.coeff(0) * (wn(1) - (wc(0) + env.const(1 << LIMB_BITS) * wn(0))))
    return out


def range_check1_constraints(env: Env):
    """RangeCheck1::constraint_checks (range_check/circuitgates.rs:279-345): 20 crumbs over the two rows + the decomposition."""
    wc, wn = env.witness_curr, env.witness_next
    out = [_crumb(env, wc(2))] + [_crumb(env, wc(i)) for i in range(7, 15)] + [_crumb(env, wn(i)) for i in range(3)] + [_crumb(env, wn(i)) for i in range(7, 15)]
    limbs = [(wn(i), 2) for i in range(14, 6, -1)] + [(wn(i), 2) for i in range(2, -1, -1)] + [(wc(i), 2) for i in range(14, 6, -1)] + \
            [(wc(i), 12) for i in range(6, 2, -1)] + [(wc(2), 2)]
    out.append(_weighted_sum(env, limbs) - wc(0))
    return out


def rot64_constraints(env: Env):
    """Rot64::constraint_checks (rot.rs:190-237); coefficient 0 = 2^rot."""
    wc, wn = env.witness_curr, env.witness_next
    out = [_crumb(env, wc(i)) for i in range(7, 15)]
    word, rotated, excess, shifted, two_rot = wc(0), wc(1), wc(2), wn(0), env.coeff(0)
    t64 = env.const(1 << 64)
    out.append(word * two_rot - (excess * t64 + shifted))
    out.append(rotated - (shifted + excess))
    limbs = [(wc(i), 2) for i in range(14, 6, -1)] + [(wc(i), 12) for i in range(6, 2, -1)]
    out.append(_weighted_sum(env, limbs) - (excess - two_rot + t64))
    return out


def foreign_field_add_constraints(env: Env):
    """ForeignFieldAdd::constraint_checks (foreign_field_add/circuitgates.rs:133-186); coefficients 0..2 = the foreign modulus' limbs, 3 = the sign."""
    wc, wn = env.witness_curr, env.witness_next
    L = env.const(1 << LIMB_BITS)
    fm, sign = [env.coeff(i) for i in range(3)], env.coeff(3)
    ovf, carry = wc(6), wc(7)
    compact = lambda lo, mi: lo + mi * L
    bot = compact(wc(0), wc(1)) + sign * compact(wc(3), wc(4)) - ovf * compact(fm[0], fm[1]) - carry * env.const(1 << (2 * LIMB_BITS))
    top = wc(2) + sign * wc(5) - ovf * fm[2] + carry
    return [ovf * (ovf - sign), carry * (carry - 1) * (carry + 1), bot - compact(wn(0), wn(1)), top - wn(2)]


def foreign_field_mul_constraints(env: Env):
    """ForeignFieldMul::constraint_checks (foreign_field_mul/circuitgates.rs:196-372); coefficient 0 = the top limb of the foreign modulus,
    1..3 = the limbs of its negation (2^264 - f)."""
    wc, wn = env.witness_curr, env.witness_next
    L, L2, L3 = env.const(1 << LIMB_BITS), env.const(1 << (2 * LIMB_BITS)), env.const(1 << (3 * LIMB_BITS))
    a, b = [wc(i) for i in range(3)], [wc(3 + i) for i in range(3)]
    c1 = [(wc(7), 12), (wc(8), 12), (wc(9), 12), (wc(10), 12), (wn(8), 12), (wn(9), 12), (wn(10), 12), (wc(11), 2), (wc(12), 2), (wc(13), 2), (wc(14), 1)]
    carry1 = _weighted_sum(env, c1)                           # 2^84 | 2^86 | 2^88 | 2^90 for the crumbs and the bit
    carry0 = wn(11)
    q = [wn(2), wn(3), wn(4)]
    rem = [wn(0), wn(1)]
    p1_lo, p1_hi0, p1_hi1 = wc(6), wn(6), wn(7)
    hi_f, nf = env.coeff(0), [env.coeff(1 + i) for i in range(3)]
    prod = [a[0] * b[0] + q[0] * nf[0],
            a[0] * b[1] + a[1] * b[0] + q[0] * nf[1] + q[1] * nf[0],
            a[0] * b[2] + a[2] * b[0] + a[1] * b[1] + q[0] * nf[2] + q[2] * nf[0] + q[1] * nf[1]]
    nat = lambda v: L2 * v[2] + L * v[1] + v[0]
    q_n = nat(q).cache()
    r_n = L2 * rem[1] + rem[0]
    bound = q[2] + L - hi_f - 1
    p1_hi = (L * p1_hi1 + p1_hi0).cache()
    return [_crumb(env, p1_hi1), _crumb(env, carry0), prod[1] - (L * p1_hi + p1_lo),
            L2 * carry0 - (prod[0] + L * p1_lo - rem[0]),
            nat(a) * nat(b) + q_n * nat(nf) - r_n - q_n * L3,
            _crumb(env, wc(11)), _crumb(env, wc(12)), _crumb(env, wc(13)), wc(14).square() - wc(14),
            L * carry1 - (prod[2] + p1_hi + carry0 - rem[1]), wn(5) - bound]


GATES = {"Poseidon": (poseidon_constraints, 15), "CompleteAdd": (complete_add_constraints, 7), "VarBaseMul": (varbasemul_constraints, 21),
         "EndoMul": (endomul_constraints, 12), "EndoMulScalar": (endomul_scalar_constraints, 11), "Xor16": (xor16_constraints, 3),
         "RangeCheck0": (range_check0_constraints, 10), "RangeCheck1": (range_check1_constraints, 21), "Rot64": (rot64_constraints, 11),
         "ForeignFieldAdd": (foreign_field_add_constraints, 4), "ForeignFieldMul": (foreign_field_mul_constraints, 11)}


# Kimchi Poseidon MDS matrices (poseidon/src/pasta/fp_kimchi.rs, fq_kimchi.rs: `mds`), by field id 0 = Fp, 1 = Fq -- the constants of
# Constant::Mds in the Poseidon gate's expression (the round constants are circuit data: the gate's coefficient columns).
POSEIDON_MDS = {
    0: [[12035446894107573964500871153637039653510326950134440362813193268448863222019, 25461374787957152039031444204194007219326765802730624564074257060397341542093, 27667907157110496066452777015908813333407980290333709698851344970789663080149], [4491931056866994439025447213644536587424785196363427220456343191847333476930, 14743631939509747387607291926699970421064627808101543132147270746750887019919, 9448400033389617131295304336481030167723486090288313334230651810071857784477], [10525578725509990281643336361904863911009900817790387635342941550657754064843, 27437632000253211280915908546961303399777448677029255413769125486614773776695, 27566319851776897085443681456689352477426926500749993803132851225169606086988]],
    1: [[28115781186772277486790024060542467295096710153315236019619365740021995624782, 22098002279041163367053200604969603243328318626084412751290336872362628294144, 10518156075882958317589806716220047551309200159506906232124952575033472931386], [8515206633865386306014865142947895502833797732365705727001733785057042819852, 19310731234716792175834594131802557577955166208124819468043130037927500684373, 361439796332338311597104753147071943681730695313819021679602959964518909239], [2193808570710678216879007026210418088296432071066284289131688133644970611483, 1201496953174589855481629688627002262719699487577300614284420648015658009380, 11619800255560837597192574795389782851917036920101027584480912719351481334717]],
}


# The two arguments every circuit has, in the gate kernels' column convention, so that tools/gen_gate_kernels.py compiles them beside
# the library (the token lists at the top of this file are the same expressions for the token machine).
def generic_expression(env: Env, alpha: int = 0):
    """index(Generic) * (c1 + alpha c2), c = q_l l + q_r r + q_o o + q_m l r + q_c on either half of the double generic gate
    (generic.rs:100-131).  Columns: witness 0..5, coefficients 15..24, the generic selector 30; constants [1, alpha]."""
    acc = None
    for g, a in ((0, env.param(1)), (1, env.param(alpha))):
        w = [env.witness_curr(3 * g + i) for i in range(3)]
        c = [env.coeff(5 * g + i) for i in range(5)]
        term = a * (c[0] * w[0] + c[1] * w[1] + c[2] * w[2] + c[3] * w[0] * w[1] + c[4])
        acc = term if acc is None else acc + term
    return env.column(30) * acc


PERM_Z_COL, PERM_X_COL, PERM_ZKPM_COL = 22, 23, 24


def permutation_expression(env: Env, gamma: int = 0, beta: int = 0, alpha0: int = 0, bshifts=(0,) * 7):
    """alpha0 zkpm(x) (z(x) prod_i (w_i + gamma + x beta shift_i) - z(x w) prod_i (w_i + gamma + sigma_i beta)) (permutation.rs:225-288).
    Columns: witness 0..6, sigma_i at 15 + i, z 22, x 23, zkpm 24; constants [gamma, beta, alpha0, beta shift_0 .. beta shift_6]."""
    g, b, a0 = env.param(gamma), env.param(beta), env.param(alpha0)
    bs = [env.param(v) for v in bshifts]
    x = env.column(PERM_X_COL)
    lhs = rhs = None
    for i in range(7):
        wg = env.witness_curr(i) + g
        wg.cached = True
        l = wg + x * bs[i]
        r = wg + env.coeff(i) * b
        lhs = l if lhs is None else lhs * l
        rhs = r if rhs is None else rhs * r
    return (lhs * env.column(PERM_Z_COL) - rhs * env.column_next(PERM_Z_COL)) * a0 * env.column(PERM_ZKPM_COL)


COMPILED_EXTRA = {"Generic": generic_expression, "Permutation": permutation_expression}


def gate_program(name: str, p: int, alpha: int, selector_col: int = 30, mds=None, endo: int = 0, w0: int = 0, c0: int = 15):
    """(tokens, constants as integers) of index(name) * combined constraints, alpha powers from alpha^0 (every gate's
    constraints start at the first of the 21 gate alphas, linearization.rs:56-58)."""
    fn, count = GATES[name]
    env = Env(p, w0=w0, c0=c0, mds=mds, endo=endo)
    cs = fn(env)
    assert len(cs) == count, (name, len(cs))
    expr = combined_constraints(env, selector_col, cs, alpha)
    toks = compile_tokens(env, expr)
    return toks, list(env.consts)


# ---------------------------------------------------------------------------------------------------------------- lookups
# The lookup argument's constraints (kimchi/src/circuits/lookup/constraints.rs:378-673, generate_feature_flags = false) as a
# token program.  The pattern data (which cells of a row are looked up, into which table) is the reference's
# LookupPattern::lookups (lookups.rs:417-487); it is protocol data, restated here (the product carries its own copy of the protocol data).
LOOKUP_XOR_TABLE_ID, LOOKUP_RANGE_CHECK_TABLE_ID = 0, 1
LOOKUP_PATTERNS = {                      # name -> list of (table id: int | ("wit", column), [witness columns of the entry])
    "Xor": [(LOOKUP_XOR_TABLE_ID, [3 + i, 7 + i, 11 + i]) for i in range(4)],
    "Lookup": [(("wit", 0), [2 * i + 1, 2 * i + 2]) for i in range(3)],
    "RangeCheck": [(LOOKUP_RANGE_CHECK_TABLE_ID, [c]) for c in range(3, 7)],
    "ForeignFieldMul": [(LOOKUP_RANGE_CHECK_TABLE_ID, [c]) for c in range(7, 11)],
}
LOOKUP_PATTERN_ORDER = ["Xor", "Lookup", "RangeCheck", "ForeignFieldMul"]


def lookup_max_per_row(patterns):
    return max(len(LOOKUP_PATTERNS[q]) for q in patterns)


def lookup_max_joint_size(patterns):
    return max(len(e) for q in patterns for _, e in LOOKUP_PATTERNS[q])


def _joint_value(env: Env, joint_combiner: Node, table_id_combiner: Node, cells, table_id):
    """combine_table_entry (tables/mod.rs:147-162): Horner in the joint combiner from the last column + table_id_combiner * id."""
    acc = None
    for c in reversed(cells):
        acc = c if acc is None else joint_combiner * acc + c
    tid = env.witness_curr(table_id[1]) if isinstance(table_id, tuple) else env.const(table_id)
    return acc + table_id_combiner * tid if acc is not None else table_id_combiner * tid


def lookup_constraints(env: Env, patterns, cols, joint_combiner: int, table_id_combiner: int, beta: int, gamma: int, dummy_value: int = 0):
    """The 3 + max_per_row (+ zero padding to 7) lookup constraints in the reference's order.  `cols`: column numbers of
    'sorted' (list of max_per_row + 1), 'aggreg', 'table', 'selector' (dict pattern -> column), and of the three row-set
    atoms 'vanish' (VanishesOnZeroKnowledgeAndPreviousRows), 'l0' (UnnormalizedLagrangeBasis(0)), 'lfinal'
    (UnnormalizedLagrangeBasis(-zk_rows - 1)), which the caller provides as evaluation columns."""
    patterns = [q for q in LOOKUP_PATTERN_ORDER if q in patterns]
    mpr = lookup_max_per_row(patterns)
    p = env.p
    jc, tic = env.const(joint_combiner), env.const(table_id_combiner)
    g, b = env.const(gamma), env.const(beta)
    gb1 = env.const(gamma * (1 + beta) % p)
    b1m = pow(1 + beta, mpr, p)

    def f_term(spec):                    # (1 + beta)^max_per_row * (gamma + dummy)^padding * prod (gamma + joint value)
        acc = env.const(pow((gamma + dummy_value) % p, mpr - len(spec), p) * b1m % p)
        for tid, entry in spec:
            acc = acc * (g + _joint_value(env, jc, tic, [env.witness_curr(c) for c in entry], tid))
        return acc

    indicator = None
    for q in patterns:
        sel = env.column(cols["selector"][q])
        indicator = sel if indicator is None else indicator + sel
    f_chunk = (env.one() - indicator) * f_term([])
    for q in patterns:
        f_chunk = f_chunk + env.column(cols["selector"][q]) * f_term(LOOKUP_PATTERNS[q])
    t_chunk = gb1 + env.column(cols["table"]) + b * env.column_next(cols["table"])
    numerator = f_chunk * t_chunk
    denominator = None
    for i in range(mpr + 1):
        c, nx = env.column(cols["sorted"][i]), env.column_next(cols["sorted"][i])
        term = (gb1 + c + b * nx) if i % 2 == 0 else (gb1 + nx + b * c)
        denominator = term if denominator is None else denominator * term
    aggreg_eq = env.column_next(cols["aggreg"]) * denominator - env.column(cols["aggreg"]) * numerator
    res = [env.column(cols["vanish"]) * aggreg_eq,
           env.column(cols["l0"]) * (env.column(cols["aggreg"]) - env.one()),
           env.column(cols["lfinal"]) * (env.column(cols["aggreg"]) - env.one())]
    for i in range(mpr):
        basis = env.column(cols["lfinal"] if i % 2 == 0 else cols["l0"])
        res.append(basis * (env.column(cols["sorted"][i]) - env.column(cols["sorted"][i + 1])))
    if cols.get("runtime") is not None:                     # runtime tables: the constraints are padded to 3 + 4, then RT(x) * selector_RT(x) (constraints.rs:658-680, runtime_tables.rs:59-66)
        res += [env.const(0)] * (4 - mpr)
        res.append(env.column(cols["runtime"]) * env.column(cols["runtime_selector"]))
    return res


def lookup_program(p: int, patterns, cols, joint_combiner: int, table_id_combiner: int, beta: int, gamma: int, alpha: int, alpha0: int = 0,
                   dummy_value: int = 0, w0: int = 0):
    """(tokens, constants) of sum_i alpha^(alpha0 + i) * lookup constraint_i (prover.rs:874-903)."""
    env = Env(p, w0=w0)
    cs = lookup_constraints(env, patterns, cols, joint_combiner, table_id_combiner, beta, gamma, dummy_value)
    return compile_tokens(env, combined_constraints(env, None, cs, alpha, alpha0)), list(env.consts)


def lookup_aggregation_programs(p: int, patterns, cols, joint_combiner: int, table_id_combiner: int, beta: int, gamma: int, dummy_value: int = 0, w0: int = 0):
    """Two token programs for the rows of the aggregation (constraints.rs:233-338): the numerator f_chunk * t_chunk and the
    denominator s_chunk of row i (cells of rows i and i + 1); aggreg[i + 1] = aggreg[i] * numerator_i / denominator_i."""
    patterns = [q for q in LOOKUP_PATTERN_ORDER if q in patterns]
    mpr = lookup_max_per_row(patterns)
    out = []
    for which in ("num", "den"):
        env = Env(p, w0=w0)
        jc, tic = env.const(joint_combiner), env.const(table_id_combiner)
        g, b = env.const(gamma), env.const(beta)
        gb1 = env.const(gamma * (1 + beta) % p)
        if which == "num":
            b1m = pow(1 + beta, mpr, p)

            def f_term(spec):
                acc = env.const(pow((gamma + dummy_value) % p, mpr - len(spec), p) * b1m % p)
                for tid, entry in spec:
                    acc = acc * (g + _joint_value(env, jc, tic, [env.witness_curr(c) for c in entry], tid))
                return acc
            indicator = None
            for q in patterns:
                sel = env.column(cols["selector"][q])
                indicator = sel if indicator is None else indicator + sel
            expr = (env.one() - indicator) * f_term([])
            for q in patterns:
                expr = expr + env.column(cols["selector"][q]) * f_term(LOOKUP_PATTERNS[q])
            expr = expr * (gb1 + env.column(cols["table"]) + b * env.column_next(cols["table"]))
        else:
            expr = None
            for i in range(mpr + 1):
                c, nx = env.column(cols["sorted"][i]), env.column_next(cols["sorted"][i])
                term = (gb1 + c + b * nx) if i % 2 == 0 else (gb1 + nx + b * c)
                expr = term if expr is None else expr * term
        out.append((compile_tokens(env, expr), list(env.consts)))
    return out
