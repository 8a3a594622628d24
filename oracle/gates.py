"""TEST INFRASTRUCTURE ONLY (see oracle/pasta.py): kimchi's gate library restated as plain per-row arithmetic -- the checker of
the token programs in proof_systems_amd/polish.py -- together with the reference's witness generators,
so that satisfied and violated witnesses can be produced without the Rust prover.

For every gate: `<gate>_row(F, curr, nxt, coeffs, ...)` returns the list of constraint VALUES of one row exactly as
`Argument::constraint_checks` orders them (a satisfied row gives all zeros), written as straight formulas, not as an
expression tree (a different shape from the product's DSL on purpose); `<gate>_witness(...)` fills witness rows the way the
reference's gen_witness / witness functions do.

  Poseidon        kimchi/src/circuits/polynomials/poseidon.rs:351-436 (constraints), :239-290 (witness), :65-80 (state order)
  CompleteAdd     complete_add.rs:103-226; witness rules from its comments (:58-93) and verify_complete_add
  VarBaseMul      varbasemul.rs:419-455, :226-277 (single_bit), :187-224 (single_bit_witness), :356-395 (witness)
  EndoMul         endosclmul.rs:475-558 (constraints), :606-690 (gen_witness)
  EndoMulScalar   endomul_scalar.rs:174-222 (constraints), :230-290 (gen_witness)

"Parity pinned by definition only": the reference holds no vectors for gate rows; what pins these is that satisfied
witnesses of the reference's OWN generators (restated here) zero every constraint, and elliptic-curve facts (the
CompleteAdd / VarBaseMul / EndoMul outputs equal the oracle's group law)."""
from typing import List, Sequence

from . import pasta as P

COLUMNS = 15
ROUND_TO_COLS = [0, 2, 3, 4, 1]


# ------------------------------------------------------------------------------------------------------------ Poseidon
def poseidon_row(F: P.Field, curr, nxt, coeffs, mds) -> List[int]:
    p = F.p
    out = []
    for r in range(5):
        src = [curr[3 * ROUND_TO_COLS[r] + k] for k in range(3)]
        sb = [pow(x, 7, p) for x in src]
        for j in range(3):
            tgt = nxt[j] if r == 4 else curr[3 * ROUND_TO_COLS[r + 1] + j]
            out.append((tgt - (coeffs[3 * r + j] + sum(mds[j][k] * sb[k] for k in range(3)))) % p)
    return out


def poseidon_witness(F: P.Field, state: Sequence[int], mds, rc, rows: int = 11):
    """rows Poseidon gate rows + the output row: (witness rows, coefficient rows).  rc: 5 * rows triples of round constants."""
    p = F.p
    w = [[0] * COLUMNS for _ in range(rows + 1)]
    co = [[0] * COLUMNS for _ in range(rows + 1)]
    st = list(state)
    for row in range(rows):
        for r in range(5):
            for k in range(3):
                w[row][3 * ROUND_TO_COLS[r] + k] = st[k]
            sb = [pow(x, 7, p) for x in st]
            c = rc[5 * row + r]
            st = [(sum(mds[j][k] * sb[k] for k in range(3)) + c[j]) % p for j in range(3)]
            for j in range(3):
                co[row][3 * r + j] = c[j]
    for k in range(3):
        w[rows][k] = st[k]
    return w, co, st


# ---------------------------------------------------------------------------------------------------------- CompleteAdd
def complete_add_row(F: P.Field, curr) -> List[int]:
    p = F.p
    x1, y1, x2, y2, x3, y3, inf, same_x, s, inf_z, x21_inv = curr[:11]
    x21, y21 = (x2 - x1) % p, (y2 - y1) % p
    return [(x21_inv * x21 - (1 - same_x)) % p,
            same_x * x21 % p,
            (same_x * (2 * s * y1 - 3 * x1 * x1) + (1 - same_x) * (x21 * s - y21)) % p,
            (x1 + x2 + x3 - s * s) % p,
            (s * (x1 - x3) - y1 - y3) % p,
            y21 * (same_x - inf) % p,
            (y21 * inf_z - inf) % p]


def complete_add_witness(F: P.Field, p1, p2) -> List[int]:
    p = F.p
    (x1, y1), (x2, y2) = p1, p2
    same_x = 1 if x1 == x2 else 0
    same_y = y1 == y2
    inf = 1 if (same_x and not same_y) else 0
    x21_inv = 0 if same_x else F.inv((x2 - x1) % p)
    s = (3 * x1 * x1 * F.inv(2 * y1 % p)) % p if same_x else (y2 - y1) * F.inv((x2 - x1) % p) % p
    inf_z = 0 if same_y else (F.inv((y2 - y1) % p) if same_x else 0)
    x3 = (s * s - x1 - x2) % p
    y3 = (s * (x1 - x3) - y1) % p
    return [x1, y1, x2, y2, x3, y3, inf, same_x, s, inf_z, x21_inv, 0, 0, 0, 0]


# ----------------------------------------------------------------------------------------------------------- VarBaseMul
def _single_bit_values(F, b, base, s1, inp, out):
    p = F.p
    b_sign = (2 * b - 1) % p
    s1_sq = s1 * s1 % p
    rx = (s1_sq - inp[0] - base[0]) % p
    t = (inp[0] - rx) % p
    u = (2 * inp[1] - t * s1) % p
    return [(b * b - b) % p,
            ((inp[0] - base[0]) * s1 - (inp[1] - b_sign * base[1])) % p,
            (u * u - t * t % p * ((out[0] - base[0] + s1_sq) % p)) % p,
            ((out[1] + inp[1]) * t - (inp[0] - out[0]) * u) % p]


def varbasemul_row(F: P.Field, curr, nxt) -> List[int]:
    p = F.p
    accs = [(curr[2], curr[3]), (curr[7], curr[8]), (curr[9], curr[10]), (curr[11], curr[12]), (curr[13], curr[14]), (nxt[0], nxt[1])]
    bits = [nxt[2 + i] for i in range(5)]
    ss = [nxt[7 + i] for i in range(5)]
    base = (curr[0], curr[1])
    acc = curr[4]
    for b in bits:
        acc = (b + 2 * acc) % p
    out = [(curr[5] - acc) % p]
    for i in range(5):
        out += _single_bit_values(F, bits[i], base, ss[i], accs[i], accs[i + 1])
    return out


def varbasemul_witness(F: P.Field, base, bits: Sequence[int], acc0):
    """witness() (varbasemul.rs:356-395): 5 bits per pair of rows.  Returns (rows, final accumulator, n)."""
    p = F.p
    assert len(bits) % 5 == 0
    rows = [[0] * COLUMNS for _ in range(2 * (len(bits) // 5))]
    acc, n_acc = acc0, 0
    acc_cols = [(2, 3), (7, 8), (9, 10), (11, 12), (13, 14)]
    for chunk in range(len(bits) // 5):
        r0, r1 = rows[2 * chunk], rows[2 * chunk + 1]
        r0[0], r0[1] = base
        r0[4] = n_acc
        for i in range(5):
            b = bits[5 * chunk + i]
            n_acc = (2 * n_acc + b) % p
            xi, yi = acc
            s1 = (yi - base[1] * ((2 * b - 1) % p)) * F.inv((xi - base[0]) % p) % p
            s1_sq = s1 * s1 % p
            s2 = (2 * yi * F.inv((2 * xi + base[0] - s1_sq) % p) - s1) % p
            ox = (base[0] + s2 * s2 - s1_sq) % p
            oy = ((xi - ox) * s2 - yi) % p
            r0[acc_cols[i][0]], r0[acc_cols[i][1]] = xi, yi
            r1[2 + i] = b; r1[7 + i] = s1
            acc = (ox, oy)
        r1[0], r1[1] = acc
        r0[5] = n_acc
    return rows, acc, n_acc


# -------------------------------------------------------------------------------------------------------------- EndoMul
def endomul_row(F: P.Field, curr, nxt, endo: int) -> List[int]:
    p = F.p
    xt, yt, inv = curr[0], curr[1], curr[2]
    xp, yp, n, xr, yr, s1, s3, b1, b2, b3, b4 = curr[4:15]
    xs, ys, n_next = nxt[4], nxt[5], nxt[6]
    xq1 = (1 + b1 * (endo - 1)) * xt % p
    xq2 = (1 + b3 * (endo - 1)) * xt % p
    yq1 = (2 * b2 - 1) * yt % p
    yq2 = (2 * b4 - 1) * yt % p
    s1s, s3s = s1 * s1 % p, s3 * s3 % p
    return [(b1 * b1 - b1) % p, (b2 * b2 - b2) % p, (b3 * b3 - b3) % p, (b4 * b4 - b4) % p,
            ((xq1 - xp) * s1 - (yq1 - yp)) % p,
            ((2 * xp - s1s + xq1) * ((xp - xr) * s1 + yr + yp) - 2 * yp * (xp - xr)) % p,
            ((yr + yp) ** 2 - (xp - xr) ** 2 * (s1s - xq1 + xr)) % p,
            ((xq2 - xr) * s3 - (yq2 - yr)) % p,
            ((2 * xr - s3s + xq2) * ((xr - xs) * s3 + ys + yr) - 2 * yr * (xr - xs)) % p,
            ((ys + yr) ** 2 - (xr - xs) ** 2 * (s3s - xq2 + xs)) % p,
            (16 * n + 8 * b1 + 4 * b2 + 2 * b3 + b4 - n_next) % p,
            ((xp - xr) * (xr - xs) * inv - 1) % p]


def endomul_witness(F: P.Field, endo: int, base, bits: Sequence[int], acc0):
    """gen_witness (endosclmul.rs:606-690): 4 bits per row (MSB first) + the closing row.  Returns (rows, acc, n)."""
    p = F.p
    assert len(bits) % 4 == 0
    nrows = len(bits) // 4
    rows = [[0] * COLUMNS for _ in range(nrows + 1)]
    acc, n_acc = acc0, 0
    xt, yt = base
    for i in range(nrows):
        b1, b2, b3, b4 = bits[4 * i:4 * i + 4]
        xp, yp = acc
        xq1 = (1 + (endo - 1) * b1) * xt % p; yq1 = (2 * b2 - 1) * yt % p
        s1 = (yq1 - yp) * F.inv((xq1 - xp) % p) % p
        s1s = s1 * s1 % p
        s2 = (2 * yp * F.inv((2 * xp + xq1 - s1s) % p) - s1) % p
        xr = (xq1 + s2 * s2 - s1s) % p
        yr = ((xp - xr) * s2 - yp) % p
        xq2 = (1 + (endo - 1) * b3) * xt % p; yq2 = (2 * b4 - 1) * yt % p
        s3 = (yq2 - yr) * F.inv((xq2 - xr) % p) % p
        s3s = s3 * s3 % p
        s4 = (2 * yr * F.inv((2 * xr + xq2 - s3s) % p) - s3) % p
        xs = (xq2 + s4 * s4 - s3s) % p
        ys = ((xr - xs) * s4 - yr) % p
        inv = F.inv((xp - xr) * (xr - xs) % p)
        rows[i][0], rows[i][1], rows[i][2] = xt, yt, inv
        rows[i][4:15] = [xp, yp, n_acc, xr, yr, s1, s3, b1, b2, b3, b4]
        acc = (xs, ys)
        n_acc = (16 * n_acc + 8 * b1 + 4 * b2 + 2 * b3 + b4) % p
    rows[nrows][4], rows[nrows][5], rows[nrows][6] = acc[0], acc[1], n_acc
    rows[nrows][0], rows[nrows][1] = xt, yt
    return rows, acc, n_acc


# -------------------------------------------------------------------------------------------------------- EndoMulScalar
def _poly(F, coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % F.p
    return acc


def endomul_scalar_row(F: P.Field, curr) -> List[int]:
    p = F.p
    n0, n8, a0, b0, a8, b8 = curr[:6]
    xs = curr[6:14]
    c_coeffs = [0, 11 * F.inv(6) % p, -5 * F.inv(2) % p, 2 * F.inv(3) % p]
    d_minus_c = [p - 1, 3, p - 1]
    crumb = [p - 6, 11, p - 6, 1]
    cs = [_poly(F, c_coeffs, x) for x in xs]
    ds = [(c + _poly(F, d_minus_c, x)) % p for c, x in zip(cs, xs)]
    n, a, b = n0, a0, b0
    for x, c, d in zip(xs, cs, ds):
        n = (4 * n + x) % p; a = (2 * a + c) % p; b = (2 * b + d) % p
    return [(n - n8) % p, (a - a8) % p, (b - b8) % p] + [_poly(F, crumb, x) * x % p for x in xs]


def endomul_scalar_witness(F: P.Field, scalar: int, endo_scalar: int, num_bits: int):
    """gen_witness (endomul_scalar.rs:230-290): 16 bits per row, MSB first.  Returns (rows, a * endo + b)."""
    p = F.p
    assert num_bits % 16 == 0
    bits_msb = [(scalar >> (num_bits - 1 - i)) & 1 for i in range(num_bits)]
    rows = []
    a, b, n = 2, 2, 0
    for r in range(num_bits // 16):
        row = [0] * COLUMNS
        row[0], row[2], row[3] = n, a, b
        for j in range(8):
            b1, b0 = bits_msb[16 * r + 2 * j], bits_msb[16 * r + 2 * j + 1]
            crumb = b0 + 2 * b1
            row[6 + j] = crumb
            a = 2 * a % p; b = 2 * b % p
            s = 1 if b0 else p - 1
            if b1:
                a = (a + s) % p
            else:
                b = (b + s) % p
            n = (4 * n + crumb) % p
        row[1], row[4], row[5] = n, a, b
        rows.append(row)
    assert n == scalar % (1 << num_bits)
    return rows, (a * endo_scalar + b) % p


# ------------------------------------------------------------------------------------------------------------ Xor16
def xor16_row(F: P.Field, curr, nxt) -> List[int]:
    """xor.rs:152-174: for in1, in2, out (columns 0, 1, 2): four 4-bit nybbles + 2^16 * the next row's value - the value."""
    p = F.p
    return [(curr[3 + 4 * i] + curr[4 + 4 * i] * 16 + curr[5 + 4 * i] * 256 + curr[6 + 4 * i] * 4096 + 65536 * nxt[i] - curr[i]) % p for i in range(3)]


def xor_witness(F: P.Field, in1: int, in2: int, bits: int) -> List[List[int]]:
    """create_xor_witness (xor.rs:262-296, layout :177-232): num_xors(bits) Xor16 rows + the zero row, as rows of 15 cells."""
    assert in1 < (1 << bits) and in2 < (1 << bits)
    out = in1 ^ in2
    rows = []
    for i in range(-(-bits // 16)):
        vals = [in1 >> (16 * i), in2 >> (16 * i), out >> (16 * i)]
        row = list(vals)
        for v in vals:
            row += [(v >> (4 * k)) & 15 for k in range(4)]
        rows.append(row)
    rows.append([0] * COLUMNS)
    return rows


def and_witness(F: P.Field, in1: int, in2: int, nbytes: int) -> List[List[int]]:
    """create_and_witness (and.rs:174-204): the xor gadget's rows + the double generic row (in1, in2, sum, sum, xor, and)."""
    rows = xor_witness(F, in1, in2, 8 * nbytes)
    s = (in1 + in2) % F.p
    rows.append([in1, in2, s, s, in1 ^ in2, in1 & in2] + [0] * 9)
    return rows


# ------------------------------------------------------------------------------------------------------------ range checks, rot, foreign field
LIMB_BITS = 88                        # KimchiForeignElement: three 88-bit limbs (foreign_field.rs)


def _crumb(p, x):
    return x * (x - 1) % p * (x - 2) % p * (x - 3) % p


def range_check0_row(F: P.Field, curr, nxt, coeffs) -> List[int]:
    """range_check/circuitgates.rs:117-163: eight crumbs (columns 7..14), the 88-bit decomposition of column 0 into six 12-bit limbs
    (columns 1..6) and the crumbs, and -- when coefficient 0 is set -- the compact form next[1] = curr[0] + 2^88 next[0]."""
    p = F.p
    out = [_crumb(p, curr[i]) for i in range(7, COLUMNS)]
    pw, acc = 1, 0
    for i in range(COLUMNS - 1, 6, -1):
        acc = (acc + pw * curr[i]) % p; pw = pw * 4 % p
    for i in range(6, 0, -1):
        acc = (acc + pw * curr[i]) % p; pw = pw * 4096 % p
    out.append((acc - curr[0]) % p)
    out.append(coeffs[0] * (nxt[1] - (curr[0] + (1 << LIMB_BITS) * nxt[0])) % p)
    return out


def range_check1_row(F: P.Field, curr, nxt) -> List[int]:
    """range_check/circuitgates.rs:279-345: 20 crumbs over the two rows and the decomposition of column 0."""
    p = F.p
    out = [_crumb(p, curr[2])] + [_crumb(p, curr[i]) for i in range(7, COLUMNS)] + [_crumb(p, nxt[i]) for i in range(3)] + [_crumb(p, nxt[i]) for i in range(7, COLUMNS)]
    pw, acc = 1, 0
    for i in range(COLUMNS - 1, 6, -1):
        acc = (acc + pw * nxt[i]) % p; pw = pw * 4 % p
    for i in range(2, -1, -1):
        acc = (acc + pw * nxt[i]) % p; pw = pw * 4 % p
    for i in range(COLUMNS - 1, 6, -1):
        acc = (acc + pw * curr[i]) % p; pw = pw * 4 % p
    for i in range(6, 2, -1):
        acc = (acc + pw * curr[i]) % p; pw = pw * 4096 % p
    acc = (acc + pw * curr[2]) % p
    out.append((acc - curr[0]) % p)
    return out


def rot64_row(F: P.Field, curr, nxt, coeffs) -> List[int]:
    """rot.rs:190-237: word * 2^rot = excess * 2^64 + shifted, rotated = shifted + excess, and the bound excess - 2^rot + 2^64 decomposed."""
    p = F.p
    out = [_crumb(p, curr[i]) for i in range(7, COLUMNS)]
    word, rotated, excess, shifted, two_rot = curr[0], curr[1], curr[2], nxt[0], coeffs[0]
    out.append((word * two_rot - (excess * (1 << 64) + shifted)) % p)
    out.append((rotated - (shifted + excess)) % p)
    pw, acc = 1, 0
    for i in range(COLUMNS - 1, 6, -1):
        acc = (acc + pw * curr[i]) % p; pw = pw * 4 % p
    for i in range(6, 2, -1):
        acc = (acc + pw * curr[i]) % p; pw = pw * 4096 % p
    out.append((acc - (excess - two_rot + (1 << 64))) % p)
    return out


def foreign_field_add_row(F: P.Field, curr, nxt, coeffs) -> List[int]:
    """foreign_field_add/circuitgates.rs:133-186: coefficients = the foreign modulus' three limbs and the sign."""
    p = F.p
    L = 1 << LIMB_BITS
    fm, sign = coeffs[:3], coeffs[3]
    llo, lmi, lhi, rlo, rmi, rhi, ovf, carry = curr[:8]
    compact = lambda lo, mi: (lo + mi * L) % p
    out = [ovf * (ovf - sign) % p, carry * (carry - 1) % p * (carry + 1) % p]
    bot = (compact(llo, lmi) + sign * compact(rlo, rmi) - ovf * compact(fm[0], fm[1]) - carry * (L * L)) % p
    top = (lhi + sign * rhi - ovf * fm[2] + carry) % p
    out.append((bot - compact(nxt[0], nxt[1])) % p)
    out.append((top - nxt[2]) % p)
    return out


def foreign_field_mul_row(F: P.Field, curr, nxt, coeffs) -> List[int]:
    """foreign_field_mul/circuitgates.rs:196-372: coefficients = the top limb of the foreign modulus, then the three limbs of its negation."""
    p = F.p
    L = 1 << LIMB_BITS
    a, b = curr[0:3], curr[3:6]
    c1 = [curr[7], curr[8], curr[9], curr[10], nxt[8], nxt[9], nxt[10], curr[11], curr[12], curr[13], curr[14]]
    carry1 = (sum(c1[i] << (12 * i) for i in range(8)) + (c1[8] << 86) + (c1[9] << 88) + (c1[10] << 90)) % p
    carry0 = nxt[11]
    q = nxt[2:5]
    q_hi_bound = nxt[5]
    rem = nxt[0:2]
    p1_lo, p1_hi0, p1_hi1 = curr[6], nxt[6], nxt[7]
    hi_f, nf = coeffs[0], coeffs[1:4]
    prod = [(a[0] * b[0] + q[0] * nf[0]) % p,
            (a[0] * b[1] + a[1] * b[0] + q[0] * nf[1] + q[1] * nf[0]) % p,
            (a[0] * b[2] + a[2] * b[0] + a[1] * b[1] + q[0] * nf[2] + q[2] * nf[0] + q[1] * nf[1]) % p]
    nat = lambda v: (L * L * v[2] + L * v[1] + v[0]) % p
    a_n, b_n, q_n, nf_n = nat(a), nat(b), nat(q), nat(nf)
    r_n = (L * L * rem[1] + rem[0]) % p
    bound = (q[2] + L - hi_f - 1) % p
    p1_hi = (L * p1_hi1 + p1_hi0) % p
    return [_crumb(p, p1_hi1), _crumb(p, carry0), (prod[1] - (L * p1_hi + p1_lo)) % p,
            (L * L * carry0 - (prod[0] + L * p1_lo - rem[0])) % p,
            (a_n * b_n + q_n * nf_n - r_n - q_n * (L * L * L)) % p,
            _crumb(p, curr[11]), _crumb(p, curr[12]), _crumb(p, curr[13]), (curr[14] * curr[14] - curr[14]) % p,
            (L * carry1 - (prod[2] + p1_hi + carry0 - rem[1])) % p, (q_hi_bound - bound) % p]


ROW_MACHINES = {"Poseidon": 15, "CompleteAdd": 7, "VarBaseMul": 21, "EndoMul": 12, "EndoMulScalar": 11, "Xor16": 3,
                "RangeCheck0": 10, "RangeCheck1": 21, "Rot64": 11, "ForeignFieldAdd": 4, "ForeignFieldMul": 11}


def combined_row(F: P.Field, name: str, curr, nxt, coeffs, alpha: int, mds=None, endo: int = 0) -> int:
    """sum_i alpha^i constraint_i of one row (without the selector)."""
    if name == "Poseidon":
        cs = poseidon_row(F, curr, nxt, coeffs, mds)
    elif name == "CompleteAdd":
        cs = complete_add_row(F, curr)
    elif name == "VarBaseMul":
        cs = varbasemul_row(F, curr, nxt)
    elif name == "EndoMul":
        cs = endomul_row(F, curr, nxt, endo)
    elif name == "Xor16":
        cs = xor16_row(F, curr, nxt)
    elif name == "EndoMulScalar":
        cs = endomul_scalar_row(F, curr)
    elif name == "RangeCheck0":
        cs = range_check0_row(F, curr, nxt, coeffs)
    elif name == "RangeCheck1":
        cs = range_check1_row(F, curr, nxt)
    elif name == "Rot64":
        cs = rot64_row(F, curr, nxt, coeffs)
    elif name == "ForeignFieldAdd":
        cs = foreign_field_add_row(F, curr, nxt, coeffs)
    elif name == "ForeignFieldMul":
        cs = foreign_field_mul_row(F, curr, nxt, coeffs)
    else:
        raise NotImplementedError(name)
    assert len(cs) == ROW_MACHINES[name]
    acc, a = 0, 1
    for c in cs:
        acc = (acc + a * c) % F.p; a = a * alpha % F.p
    return acc
