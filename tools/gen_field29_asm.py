#!/usr/bin/env python3
"""Generates proof_systems_amd/csrc/field29_asm.inc: Montgomery product and squaring for the Pasta
primes on NINE 29-bit limbs (R' = 2^261), one inline-asm block each, and checks the generated
instruction streams against big-integer arithmetic with a small interpreter (run: python3
tools/gen_field29_asm.py --check).

Why nine 29-bit limbs: a 29 x 29 product is 58 bits, so a whole column of the product scan
(9 limb products + 5 reduction products + the carry) fits ONE 64-bit accumulator -- every limb
product is a single v_mad_u64_u32, no carry word, no v_addc.  The eight-limb 32-bit schedule of
field_mulasm.inc needs 104 MADs + 104 v_addc (254 instructions); this one needs 126 MADs + 61.

    p = 2^254 + t 2^32 + 1 = [1, P1, P2, P3, P4, 0, 0, 0, 2^22] in 29-bit limbs, so -1/p = -1 mod 2^29.
    Every reduction column k carries an offset of MASK = 2^29 - 1:  with C' = C + MASK,
        m_k = -C mod 2^29 = ~C' & MASK                 (one v_bfi_b32)
        carry = (C + m_k) >> 29 = ceil(C / 2^29) = C' >> 29   (the product m_k * p_0 is never formed)
    column k:  MADs a_i b_(k-i), MADs m_(k-l) P_l (l = 1..4), MAD m_(k-8) 2^22, MAD MASK * 1,
               m_k = bfi, acc >>= 29 (v_alignbit + v_lshrrev)                       k = 0..8
               out_(k-9) = acc & MASK, acc >>= 29                                   k = 9..16, out_8 = acc
    Result = (a b + m p) / 2^261 < a b / 2^261 + p, limbs normalised (< 2^29, the top limb takes the rest).
    Accumulator bound: 9 2^(A+B) + 4 2^58 + 2^51 + 2^36 < 2^64  <=>  A + B <= 60.7 for limbs a_i < 2^A, b_j < 2^B.

Operands (mul):  %0-%8 r (early clobber; r_k holds m_k until column k+8), %9-%17 a, %18-%26 b,
                 %27-%30 P1..P4, %31 2^22, %32 MASK (SGPRs).  Clobbers vcc, v2, v3 (the 64-bit accumulator).
Operands (sqr):  %0-%8 r, %9-%16 2*a_1..2*a_8 (scratch outputs), %17-%25 a, %26-%29 P1..P4, %30 2^22, %31 MASK.
"""
import os
import random
import re
import sys

MASK = (1 << 29) - 1
P_FP = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
P_FQ = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001


def limbs29(x, n=9):
    return [(x >> (29 * i)) & MASK for i in range(n - 1)] + [x >> (29 * (n - 1))]


# Operand maps.  Outputs first (inline-asm numbering): %0-%8 r (early clobber; r_k holds m_k until column k+8).
#   mul    : %9-%17 a, %18-%26 b, %27-%30 P1..P4, %31 2^22, %32 MASK (SGPRs), %33 2^29+1 (VGPR)
#   sqr    : %9-%16 2*a_1..2*a_8 (scratch outputs), %17-%25 a, %26-%29 P1..P4, %30 2^22, %31 MASK, %32 2^29+1
#   muladd : r = (a b + c d) / R':  %9-%17 a, %18-%26 b, %27-%35 c, %36-%44 d, %45-%48 P1..P4, %49 2^22, %50 MASK, %51 2^29+1
def gen_mul(shift64=True):
    r = lambda i: f"%{i}"
    a = lambda i: f"%{9 + i}"
    b = lambda i: f"%{18 + i}"
    P = {1: "%27", 2: "%28", 3: "%29", 4: "%30", 8: "%31"}
    prods = [(i + j, a(i), b(j)) for i in range(9) for j in range(9)]
    return _gen(r, prods, P, "%32", "%33", [], shift64)


def gen_sqr(shift64=True):
    r = lambda i: f"%{i}"
    d = lambda j: f"%{8 + j}"            # 2 * a_j, j = 1..8 -> %9..%16 (scratch OUTPUT operands)
    a = lambda i: f"%{17 + i}"
    P = {1: "%26", 2: "%27", 3: "%28", 4: "%29", 8: "%30"}
    pre = [f"v_lshlrev_b32 {d(j)}, 1, {a(j)}" for j in range(1, 9)]
    prods = [(i + j, a(i), a(j) if i == j else d(j)) for i in range(9) for j in range(i, 9)]
    return _gen(r, prods, P, "%31", "%32", pre, shift64)


def gen_muladd(shift64=True):
    r = lambda i: f"%{i}"
    a = lambda i: f"%{9 + i}"
    b = lambda i: f"%{18 + i}"
    c = lambda i: f"%{27 + i}"
    d = lambda i: f"%{36 + i}"
    P = {1: "%45", 2: "%46", 3: "%47", 4: "%48", 8: "%49"}
    prods = [(i + j, a(i), b(j)) for i in range(9) for j in range(9)] + [(i + j, c(i), d(j)) for i in range(9) for j in range(9)]
    return _gen(r, prods, P, "%50", "%51", [], shift64)


ACC, LO, HI = "v[2:3]", "v2", "v3"            # the 64-bit column accumulator: a fixed aligned pair (clobbered)


def _gen(r, prods, P, mask, pairk, pre, shift64):
    """prods: (column, x, y) limb products.  The MASK offsets of two neighbouring reduction columns are added by ONE MAD:
    column k (even) receives MASK + MASK 2^29 = (2^29 - 1)(2^29 + 1) -- the second term is column k+1's offset, 29 bits
    up, invisible to m_k and delivered by the shift; column 8 adds its own MASK alone."""
    L = list(pre)
    acc, lo, hi = ACC, LO, HI
    by_col = {}
    for (k, x, y) in prods:
        by_col.setdefault(k, []).append((x, y))
    first = True
    for k in range(17):
        for (x, y) in sorted(by_col.get(k, [])):
            L.append(f"v_mad_u64_u32 {acc}, vcc, {x}, {y}, {'0' if first else acc}")
            first = False
        for l in (1, 2, 3, 4, 8):
            if 0 <= k - l <= 8:
                L.append(f"v_mad_u64_u32 {acc}, vcc, {r(k - l)}, {P[l]}, {acc}")
        if k <= 8:
            if k == 8:
                L.append(f"v_mad_u64_u32 {acc}, vcc, {mask}, 1, {acc}")
            elif k % 2 == 0:
                L.append(f"v_mad_u64_u32 {acc}, vcc, {mask}, {pairk}, {acc}")
            L.append(f"v_bfi_b32 {r(k)}, {lo}, 0, {mask}")            # m_k = ~lo & MASK
        else:
            L.append(f"v_and_b32 {r(k - 9)}, {mask}, {lo}")
        if shift64:
            L.append(f"v_lshrrev_b64 {acc}, 29, {acc}")
        else:
            L.append(f"v_alignbit_b32 {lo}, {hi}, {lo}, 29")
            if k < 16:
                L.append(f"v_lshrrev_b32 {hi}, 29, {hi}")
    L.append(f"v_mov_b32 {r(8)}, {lo}")
    return L


# ------------------------------------------------------------------------------------------- interpreter
def simulate(lines, regs):
    """regs: dict name -> int (32-bit values; the accumulator is two entries '<acc>_lo' / '<acc>_hi')."""
    def val(tok):
        tok = tok.strip()
        if tok in regs:
            return regs[tok]
        if re.fullmatch(r"-?\d+", tok):
            return int(tok) & 0xffffffff
        if tok.startswith("0x"):
            return int(tok, 16)
        raise KeyError(tok)

    def val64(tok):
        tok = tok.strip()
        if tok == "0":
            return 0
        assert tok == ACC
        return regs[LO] | (regs[HI] << 32)

    for ln in lines:
        op, rest = ln.split(None, 1)
        args = [x.strip() for x in rest.split(",")]
        if op == "v_mad_u64_u32":
            d, _vcc, x, y, c = args
            v = val(x) * val(y) + val64(c)
            assert v < (1 << 64), "64-bit accumulator overflow in: " + ln
            regs[LO], regs[HI] = v & 0xffffffff, v >> 32
        elif op == "v_bfi_b32":
            d, s0, s1, s2 = args
            regs[d] = (val(s0) & val(s1)) | (~val(s0) & val(s2) & 0xffffffff)
        elif op == "v_and_b32":
            d, x, y = args
            regs[d] = val(x) & val(y)
        elif op == "v_alignbit_b32":
            d, h, l, sh = args
            regs[d] = (((val(h) << 32) | val(l)) >> val(sh)) & 0xffffffff
        elif op == "v_lshrrev_b32":
            d, sh, x = args
            regs[d] = val(x) >> val(sh)
        elif op == "v_lshlrev_b32":
            d, sh, x = args
            regs[d] = (val(x) << val(sh)) & 0xffffffff
        elif op == "v_lshrrev_b64":
            d, sh, x = args
            v = val64(x) >> val(sh)
            regs[LO], regs[HI] = v & 0xffffffff, v >> 32
        elif op == "v_mov_b32":
            d, x = args
            regs[d] = val(x)
        else:
            raise ValueError(op)
    return regs


def check(p, name, trials=300):
    rnd = random.Random(29)
    pl = limbs29(p)
    assert pl[0] == 1 and pl[5] == pl[6] == pl[7] == 0 and pl[8] == 1 << 22
    Rinv = pow(1 << 261, -1, p)
    for shift64 in (False, True):
        mul, sqr, mad = gen_mul(shift64), gen_sqr(shift64), gen_muladd(shift64)
        for t in range(trials):
            # operands as the accumulation kernel produces them: one normalised, one with limbs up to 2^31 - 1 (A + B <= 60)
            kind = t % 4
            if kind == 0:
                al = limbs29(rnd.randrange(0, 32 * p)); bl = limbs29(rnd.randrange(0, 2 * p))
            elif kind == 1:
                al = [MASK] * 8 + [(1 << 29) - 1]; bl = [(1 << 31) - 1] * 9           # worst-case magnitudes
            elif kind == 2:
                al = [rnd.randrange(0, 1 << 30) for _ in range(9)]; bl = [rnd.randrange(0, 1 << 30) for _ in range(9)]
            else:
                al = limbs29(rnd.choice([0, 1, p - 1, p, p + 1, 2 * p - 1])); bl = limbs29(rnd.choice([0, 1, p - 1, p, 2 * p]))
            av = sum(x << (29 * i) for i, x in enumerate(al)); bv = sum(x << (29 * i) for i, x in enumerate(bl))
            regs = {f"%{i}": 0 for i in range(9)}
            regs.update({f"%{9 + i}": al[i] for i in range(9)})
            regs.update({f"%{18 + i}": bl[i] for i in range(9)})
            regs.update({"%27": pl[1], "%28": pl[2], "%29": pl[3], "%30": pl[4], "%31": 1 << 22, "%32": MASK, "%33": (1 << 29) + 1, LO: 0, HI: 0})
            simulate(mul, regs)
            out = [regs[f"%{i}"] for i in range(9)]
            ov = sum(x << (29 * i) for i, x in enumerate(out))
            assert all(x <= MASK for x in out[:8]), "limbs not normalised"
            assert ov % p == av * bv * Rinv % p, f"{name} mul mismatch"
            assert ov < av * bv // (1 << 261) + p + 1, "bound"
            # squaring: limbs < 2^30 (doubled operand < 2^31)
            if kind in (0, 3):
                sl = limbs29(av % (16 * p))
            else:
                sl = [rnd.randrange(0, 1 << 30) for _ in range(9)]
            sv = sum(x << (29 * i) for i, x in enumerate(sl))
            regs = {f"%{i}": 0 for i in range(9)}
            regs.update({f"%{17 + i}": sl[i] for i in range(9)})
            regs.update({"%26": pl[1], "%27": pl[2], "%28": pl[3], "%29": pl[4], "%30": 1 << 22, "%31": MASK, "%32": (1 << 29) + 1, LO: 0, HI: 0})
            regs.update({f"%{8 + j}": 0 for j in range(1, 9)})
            simulate(sqr, regs)
            out = [regs[f"%{i}"] for i in range(9)]
            ov = sum(x << (29 * i) for i, x in enumerate(out))
            assert ov % p == sv * sv * Rinv % p, f"{name} sqr mismatch"
            assert all(x <= MASK for x in out[:8])
            # fused a b + c d: one normalised and one un-normalised operand per product (the r.y step of madd29);
            # worst-case magnitudes in kind 1: limbs 2^29 - 1 against 1.5 * 2^30 and 1.25 * 2^30
            if kind == 1:
                cl = [MASK] * 9; dl = [(5 << 28) - 1] * 9; bl2 = [(3 << 29) - 1] * 9; al2 = [MASK] * 9
            else:
                al2, bl2 = al, [min(x, (3 << 29) - 1) for x in bl]
                cl = limbs29(rnd.randrange(0, 2 * p)); dl = [rnd.randrange(0, 5 << 28) for _ in range(9)]
            regs = {f"%{i}": 0 for i in range(9)}
            for base, vec in ((9, al2), (18, bl2), (27, cl), (36, dl)):
                regs.update({f"%{base + i}": vec[i] for i in range(9)})
            regs.update({"%45": pl[1], "%46": pl[2], "%47": pl[3], "%48": pl[4], "%49": 1 << 22, "%50": MASK, "%51": (1 << 29) + 1, LO: 0, HI: 0})
            simulate(mad, regs)
            out = [regs[f"%{i}"] for i in range(9)]
            ov = sum(x << (29 * i) for i, x in enumerate(out))
            v4 = [sum(x << (29 * i) for i, x in enumerate(v)) for v in (al2, bl2, cl, dl)]
            assert ov % p == (v4[0] * v4[1] + v4[2] * v4[3]) * Rinv % p, f"{name} muladd mismatch"
            assert all(x <= MASK for x in out[:8]) and ov < (v4[0] * v4[1] + v4[2] * v4[3]) // (1 << 261) + p + 1
    print(f"{name}: mul {len(gen_mul())} instr ({len(gen_mul(False))} with v_alignbit/v_lshrrev_b32 shifts), sqr {len(gen_sqr())}, muladd {len(gen_muladd())} -- {trials} x 2 trials OK")


# ------------------------------------------------------------------------------------------- constants + madd model
# Subtraction a - b + K p on limbs without borrows: add the limbs of K p "spread" so that every limb dominates the
# subtrahend's: C_0 = d_0 + J 2^29, C_i = d_i + J 2^29 - J (0 < i < 8), C_8 = d_8 - J  (d = limbs of K p; sum C_i 2^(29 i) = K p).
# Valid when b_i <= J MASK for i < 8 and b_8 <= d_8 - J.
SPREADS = [(7, 1), (5, 1), (4, 4), (6, 1)]        # (K, J) pairs used by madd29 (field29.cuh)


def spread(p, K, J):
    d = limbs29(K * p)
    c = [d[0] + (J << 29)] + [d[i] + (J << 29) - J for i in range(1, 8)] + [d[8] - J]
    assert sum(x << (29 * i) for i, x in enumerate(c)) == K * p and all(0 <= x < (1 << 32) for x in c)
    return c


def field_consts(p):
    R, R29 = 1 << 256, 1 << 261
    return {"P": limbs29(p),
            "ONE": limbs29(R29 % p),                        # 1 in R'-Montgomery form
            "KIN": limbs29(R29 * R29 * pow(R, -1, p) % p),  # mul29(X, KIN) = x R'  for X = x R  (canonical wire form)
            "KOUT": limbs29(R % p),                         # mul29(v, KOUT) = x R   for v = x R'
            **{f"S{K}{J}": spread(p, K, J) for (K, J) in SPREADS}}


class Madd29Model:
    """Limb-exact model of madd29 (field29.cuh): every u32 operation asserts 0 <= result < 2^32, every product goes through
    the generated instruction stream.  Returns None when an exception filter fires (the kernel then hands the task to the
    exact 32-bit path)."""

    def __init__(self, p):
        self.p = p; self.c = field_consts(p); self.mul_l, self.sqr_l, self.mad_l = gen_mul(), gen_sqr(), gen_muladd()
        self.pl = limbs29(p)

    def mul(self, a, b):
        assert max(a) * max(b) < (1 << 61) * 1.6, "limb magnitudes beyond the accumulator bound"
        regs = {f"%{i}": 0 for i in range(9)}
        regs.update({f"%{9 + i}": a[i] for i in range(9)}); regs.update({f"%{18 + i}": b[i] for i in range(9)})
        pl = self.pl
        regs.update({"%27": pl[1], "%28": pl[2], "%29": pl[3], "%30": pl[4], "%31": 1 << 22, "%32": MASK, "%33": (1 << 29) + 1, LO: 0, HI: 0})
        simulate(self.mul_l, regs)
        return [regs[f"%{i}"] for i in range(9)]

    def sqr(self, a):
        regs = {f"%{i}": 0 for i in range(9)}
        regs.update({f"%{17 + i}": a[i] for i in range(9)})
        pl = self.pl
        regs.update({"%26": pl[1], "%27": pl[2], "%28": pl[3], "%29": pl[4], "%30": 1 << 22, "%31": MASK, "%32": (1 << 29) + 1, LO: 0, HI: 0})
        regs.update({f"%{8 + j}": 0 for j in range(1, 9)})
        simulate(self.sqr_l, regs)
        return [regs[f"%{i}"] for i in range(9)]

    def muladd(self, a, b, c, d):
        regs = {f"%{i}": 0 for i in range(9)}
        for base, vec in ((9, a), (18, b), (27, c), (36, d)):
            regs.update({f"%{base + i}": vec[i] for i in range(9)})
        pl = self.pl
        regs.update({"%45": pl[1], "%46": pl[2], "%47": pl[3], "%48": pl[4], "%49": 1 << 22, "%50": MASK, "%51": (1 << 29) + 1, LO: 0, HI: 0})
        simulate(self.mad_l, regs)
        return [regs[f"%{i}"] for i in range(9)]

    @staticmethod
    def u32(x):
        assert 0 <= x < (1 << 32), f"u32 range violated: {x}"
        return x

    def subn(self, a, b, C):                      # normalised a - b + C
        r, carry = [], 0
        for i in range(9):
            t = self.u32(self.u32(a[i] + C[i] + carry) - b[i])
            if i < 8:
                r.append(t & MASK); carry = t >> 29
            else:
                r.append(t)
        return r

    def madd(self, acc, px, py):
        """acc = (x, y, zz, zzz) limb lists (x < 6p, y < 4p normalised; zz, zzz products); px, py = 32 X, 32 Y normalised."""
        c = self.c
        x, y, zz, zzz = acc
        if zz[0] <= 1:
            return None                            # possibly the identity
        U2 = self.mul(px, zz); S2 = self.mul(py, zzz)
        P = self.subn(U2, x, c["S71"]); R = self.subn(S2, y, c["S51"])
        if P[0] <= 8:
            return None                            # possibly P == 0 (mod p): equal or opposite points
        PP = self.sqr(P); PPP = self.mul(P, PP); Q = self.mul(x, PP); RR = self.sqr(R)
        sub = [self.u32(PPP[i] + 2 * Q[i]) for i in range(9)]
        rx = self.subn(RR, sub, c["S44"])
        t = [self.u32(self.u32(Q[i] + c["S61"][i]) - rx[i]) for i in range(9)]          # not normalised: limbs < 2^29 + 2^30
        yn = [self.u32(c["S51"][i] - y[i]) for i in range(9)]                            # 5p - y, not normalised
        ry = self.muladd(R, t, yn, PPP)                                                  # R (Q - rx) - y PPP with ONE reduction
        return (rx, ry, self.mul(zz, PP), self.mul(zzz, PPP))

    def add(self, A, B):
        """Full XYZZ addition of two lazy points (add29, field29.cuh): both operands as madd leaves an accumulator (x < 6p, y < 4p, zz, zzz < 2p,
        normalised).  None when P = U2 - U1 = 0 (mod p) cannot be excluded (equal or opposite points: the exact path decides)."""
        c = self.c
        x1, y1, zz1, zzz1 = A
        x2, y2, zz2, zzz2 = B
        U1 = self.mul(x1, zz2); U2 = self.mul(x2, zz1); S1 = self.mul(y1, zzz2); S2 = self.mul(y2, zzz1)
        P = self.subn(U2, U1, c["S51"]); R = self.subn(S2, S1, c["S51"])
        if P[0] <= 8:
            return None
        PP = self.sqr(P); PPP = self.mul(P, PP); Q = self.mul(U1, PP); RR = self.sqr(R)
        sub = [self.u32(PPP[i] + 2 * Q[i]) for i in range(9)]
        rx = self.subn(RR, sub, c["S44"])
        t = [self.u32(self.u32(Q[i] + c["S61"][i]) - rx[i]) for i in range(9)]
        yn = [self.u32(c["S51"][i] - S1[i]) for i in range(9)]
        ry = self.muladd(R, t, yn, PPP)
        return (rx, ry, self.mul(self.mul(zz1, zz2), PP), self.mul(self.mul(zzz1, zzz2), PPP))


def val29(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def check_madd(p, name, chains=12, length=40):
    """Random chains of mixed additions through the limb model against affine big-integer arithmetic, plus value bounds."""
    rnd = random.Random(7)
    M = Madd29Model(p)
    R29inv = pow(1 << 261, -1, p)
    b = 5

    def rand_point():
        while True:
            x = rnd.randrange(p); y2 = (x * x * x + b) % p
            if pow(y2, (p - 1) // 2, p) == 1:
                # p = 1 mod 2^32: no shortcut square root; Tonelli-Shanks
                q, s = p - 1, 0
                while q % 2 == 0: q //= 2; s += 1
                z = 5
                while pow(z, (p - 1) // 2, p) != p - 1: z += 1
                m_, c_, t_, r_ = s, pow(z, q, p), pow(y2, q, p), pow(y2, (q + 1) // 2, p)
                while t_ != 1:
                    i, tt = 0, t_
                    while tt != 1: tt = tt * tt % p; i += 1
                    bb = pow(c_, 1 << (m_ - i - 1), p); m_, c_ = i, bb * bb % p; t_, r_ = t_ * c_ % p, r_ * bb % p
                return x, r_

    def aff_add(A, B):
        (x1, y1), (x2, y2) = A, B
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    maxv = [0, 0, 0, 0]
    R = 1 << 256
    for _ in range(chains):
        A = rand_point()
        X, Y = A[0] * R % p, A[1] * R % p
        acc = (M.mul(limbs29(X), M.c["KIN"]), M.mul(limbs29(Y), M.c["KIN"]), list(M.c["ONE"]), list(M.c["ONE"]))
        for _ in range(length):
            B = rand_point()
            if rnd.random() < 0.3: B = (B[0], (p - B[1]) % p)
            px, py = limbs29((B[0] * R % p) << 5), limbs29((B[1] * R % p) << 5)
            nxt = M.madd(acc, px, py)
            assert nxt is not None, "filter fired on random points (probability ~2^-25)"
            acc = nxt; A = aff_add(A, B)
            vals = [val29(v) for v in acc]
            maxv = [max(m, v / p) for m, v in zip(maxv, vals)]
            assert vals[0] < 6 * p and vals[1] < 4 * p and vals[2] < 2 * p and vals[3] < 2 * p
            assert all(l <= MASK for v in acc for l in v[:8])
            xs, ys, zzs, zzzs = [v * R29inv % p for v in vals]
            assert xs * pow(zzs, -1, p) % p == A[0] and ys * pow(zzzs, -1, p) % p == A[1], f"{name}: madd29 model mismatch"
    # filters: P = k p must be caught, zz in {0, p} must be caught
    for k in range(9):
        assert limbs29(k * p)[0] == k
    print(f"{name}: madd29 limb model OK over {chains} x {length} additions; max value / p = " + ", ".join(f"{m:.2f}" for m in maxv))


def check_add(p, name, trees=6, leaves=24):
    """add29 through the limb model: random sums of sums (operands that are themselves results of madd / add chains) against affine arithmetic."""
    rnd = random.Random(11)
    M = Madd29Model(p)
    R29inv = pow(1 << 261, -1, p)
    R = 1 << 256

    def sqrt(a):
        q, s = p - 1, 0
        while q % 2 == 0: q //= 2; s += 1
        z = 5
        while pow(z, (p - 1) // 2, p) != p - 1: z += 1
        m_, c_, t_, r_ = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
        while t_ != 1:
            i, tt = 0, t_
            while tt != 1: tt = tt * tt % p; i += 1
            bb = pow(c_, 1 << (m_ - i - 1), p); m_, c_ = i, bb * bb % p; t_, r_ = t_ * c_ % p, r_ * bb % p
        return r_

    def rand_point():
        while True:
            x = rnd.randrange(p); y2 = (x * x * x + 5) % p
            if pow(y2, (p - 1) // 2, p) == 1:
                return x, sqrt(y2)

    def aff_add(A, B):
        (x1, y1), (x2, y2) = A, B
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    def lazy(A):                                   # an affine point as the accumulation kernel starts a task (to29 of the wire form, zz = zzz = 1)
        X, Y = A[0] * R % p, A[1] * R % p
        return (M.mul(limbs29(X), M.c["KIN"]), M.mul(limbs29(Y), M.c["KIN"]), list(M.c["ONE"]), list(M.c["ONE"]))

    maxv = [0, 0, 0, 0]
    for _ in range(trees):
        items = []
        for _ in range(leaves):
            A = rand_point(); acc = lazy(A)
            for _ in range(rnd.randrange(0, 3)):   # some leaves are short madd chains (a bucket of 1-3 entries)
                B = rand_point()
                px, py = limbs29((B[0] * R % p) << 5), limbs29((B[1] * R % p) << 5)
                acc = M.madd(acc, px, py); A = aff_add(A, B)
                assert acc is not None
            items.append((acc, A))
        while len(items) > 1:                      # sequential chain and pairwise tree mixed: every operand shape add29 meets
            i = rnd.randrange(len(items) - 1)
            (a, A), (b, B) = items[i], items[i + 1]
            r = M.add(a, b)
            assert r is not None, "filter fired on random points"
            vals = [val29(v) for v in r]
            maxv = [max(m, v / p) for m, v in zip(maxv, vals)]
            assert vals[0] < 6 * p and vals[1] < 4 * p and vals[2] < 2 * p and vals[3] < 2 * p
            assert all(l <= MASK for v in r for l in v[:8])
            C = aff_add(A, B)
            xs, ys, zzs, zzzs = [v * R29inv % p for v in vals]
            assert xs * pow(zzs, -1, p) % p == C[0] and ys * pow(zzzs, -1, p) % p == C[1], f"{name}: add29 model mismatch"
            items[i:i + 2] = [(r, C)]
    # the filter: A + A and A + (-A) must be caught (P = U2 - U1 + 5p = 5p exactly when U1 = U2 as integers; 4p / 6p otherwise)
    A = rand_point(); a = lazy(A)
    B = rand_point(); px, py = limbs29((B[0] * R % p) << 5), limbs29((B[1] * R % p) << 5)
    a2 = M.madd(lazy(A), px, py)                   # A + B through one route ...
    b2 = M.madd(lazy(B), limbs29((A[0] * R % p) << 5), limbs29((A[1] * R % p) << 5))     # ... and B + A through another: same point, different representation
    assert M.add(a, a) is None and M.add(a2, b2) is None
    neg = (b2[0], limbs29((5 * p - val29(b2[1])) % (8 * p)), b2[2], b2[3])
    assert M.add(a2, neg) is None
    print(f"{name}: add29 limb model OK over {trees} trees of {leaves} leaves; max value / p = " + ", ".join(f"{m:.2f}" for m in maxv))


def emit_consts():
    out = ["// GENERATED by tools/gen_field29_asm.py -- 29-bit-limb constants of the two Pasta primes (see that file)."]
    for name, p in (("Fp29C", P_FP), ("Fq29C", P_FQ)):
        c = field_consts(p)
        out.append(f"struct {name} {{")
        out.append(f"    static constexpr u32 P1 = 0x{c['P'][1]:08x}u, P2 = 0x{c['P'][2]:08x}u, P3 = 0x{c['P'][3]:08x}u, P4 = 0x{c['P'][4]:08x}u;")
        for key in [k for k in c if k != "P"]:
            arr = ", ".join(f"0x{x:08x}u" for x in c[key])
            out.append(f"    __device__ static constexpr u32 {key.lower()}(int i) {{ constexpr u32 T[9] = {{{arr}}}; return T[i]; }}")
        out.append("};")
    return "\n".join(out) + "\n"


# ------------------------------------------------------------------------------------------- emit
def emit(name, lines):
    out = list(lines)
    return f"#define {name} \\\n    \"" + "\\n\\t\" \\\n    \"".join(out) + "\"\n"


INC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "proof_systems_amd", "csrc", "field29_asm.inc")


def render():
    mul, sqr, mad = gen_mul(), gen_sqr(), gen_muladd()
    mul32 = gen_mul(False)
    nm = lambda l: sum('v_mad' in x for x in l)
    out = ["// GENERATED by tools/gen_field29_asm.py -- do not edit.  See that file for the schedule and the bounds.",
           f"// Montgomery product on nine 29-bit limbs (R' = 2^261): {len(mul)} instructions, {nm(mul)} v_mad_u64_u32.",
           emit("KH29_MUL_ASM", mul),
           f"// Montgomery squaring: {len(sqr)} instructions, {nm(sqr)} v_mad_u64_u32.",
           emit("KH29_SQR_ASM", sqr),
           f"// (a b + c d) / R' with one reduction: {len(mad)} instructions, {nm(mad)} v_mad_u64_u32.",
           emit("KH29_MULADD_ASM", mad),
           f"// the product with v_alignbit_b32 + v_lshrrev_b32 column shifts ({len(mul32)} instructions): slower (tools/microbench.hip measures both)",
           emit("KH29_MUL_ASM_B32", mul32), emit_consts()]
    return "\n".join(out)


def main():
    if "--check" in sys.argv:
        check(P_FP, "Fp"); check(P_FQ, "Fq")
        check_madd(P_FP, "Fp"); check_madd(P_FQ, "Fq")
        check_add(P_FP, "Fp"); check_add(P_FQ, "Fq")
    text = render()
    if not os.path.exists(INC_PATH) or open(INC_PATH).read() != text:      # leave the mtime alone when nothing changed
        open(INC_PATH, "w").write(text)
    print(len(gen_mul()), "instructions per product,", len(gen_sqr()), "per squaring")


if __name__ == "__main__":
    main()
