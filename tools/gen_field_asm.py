#!/usr/bin/env python3
"""Generates proof_systems_amd/csrc/field_mulasm.inc: the hand-scheduled Montgomery product for
the Pasta primes as ONE inline-asm block (no compiler glue between columns).

Product scanning over eight 32-bit limbs.  Column accumulator = 96 bits: a 64-bit VGPR pair that
v_mad_u64_u32 accumulates into (carry-out -> VCC) + a third word collecting the carries with
v_addc_co_u32.  Two fixed pairs A = v[2:3], B = v[4:5] alternate so that moving to the next
column costs one v_mov (the pair alignment rule forbids using {hi(A), third} directly):
    column on pair X with third word hi(Y):  ... MACs ... ; out = lo(X) ; lo(Y) <- hi(X)
p = [1, P1, P2, P3, 0, 0, 0, 2^30]: m_k = -lo (one v_sub), m_k*p0 is a MAC with the inline
constant 1, m_k*2^30 a MAC with the inline constant 2.0 (= 0x40000000).
Operands: %0-%7 result, %8-%15 a, %16-%23 b, %24-%26 P1..P3 (VGPRs, also used by the final
conditional subtraction).  Clobbers: vcc, v2-v13 (accumulators + the eight m digits).
"""
import os

A = ("v[2:3]", "v2", "v3")
B = ("v[4:5]", "v4", "v5")
M = [f"v{6 + k}" for k in range(8)]


def a(i): return f"%{8 + i}"
def b(i): return f"%{16 + i}"
def t(i): return f"%{i}"
P = {1: "%24", 2: "%25", 3: "%26"}


def gen_mul(square=False):
    return _gen(A, B, M, a, b, t, P, square)


def _gen(A, B, M, a, b, t, P, square):
    L = []
    X, Y = A, B
    for k in range(15):
        prods = []
        for i in range(8):
            j = k - i
            if 0 <= j < 8:
                prods.append((a(i), b(j) if not square else a(j)))
        for pj in (1, 2, 3):
            if 0 <= k - pj < 8:
                prods.append((M[k - pj], P[pj]))
        if 0 <= k - 7 < 8:
            prods.append((M[k - 7], "2.0"))
        first = True
        for (x, y) in prods:
            if k == 0 and first:
                L.append(f"v_mad_u64_u32 {X[0]}, vcc, {x}, {y}, 0")
            else:
                L.append(f"v_mad_u64_u32 {X[0]}, vcc, {x}, {y}, {X[0]}")
            L.append(f"v_addc_co_u32 {Y[2]}, vcc, 0, {'0' if first else Y[2]}, vcc")
            first = False
        if k < 8:
            L.append(f"v_sub_u32 {M[k]}, 0, {X[1]}")
            L.append(f"v_mad_u64_u32 {X[0]}, vcc, {M[k]}, 1, {X[0]}")
            L.append(f"v_addc_co_u32 {Y[2]}, vcc, 0, {Y[2]}, vcc")
        else:
            L.append(f"v_mov_b32 {t(k - 8)}, {X[1]}")
        if k < 14:
            L.append(f"v_mov_b32 {Y[1]}, {X[2]}")
            X, Y = Y, X
        else:
            L.append(f"v_mov_b32 {t(7)}, {X[2]}")
    # conditional subtraction of p: s = t - p in M[0..7], keep t if it borrowed
    L.append(f"v_subrev_co_u32 {M[0]}, vcc, 1, {t(0)}")
    for i in (1, 2, 3):
        L.append(f"v_subb_co_u32 {M[i]}, vcc, {t(i)}, {P[i]}, vcc")
    for i in (4, 5, 6):
        L.append(f"v_subbrev_co_u32 {M[i]}, vcc, 0, {t(i)}, vcc")
    L.append(f"v_subbrev_co_u32 {M[7]}, vcc, 2.0, {t(7)}, vcc")
    for i in range(8):
        L.append(f"v_cndmask_b32 {t(i)}, {M[i]}, {t(i)}, vcc")
    return L


def emit(name, lines):
    body = "\\n\\t\"\n    \"".join(lines)
    return f'#define {name} \\\n    "' + "\\n\\t\" \\\n    \"".join(lines) + '"\n'


def gen_sqr():
    """r = a^2 * 2^-256 mod p with 36 instead of 64 limb products:
         a^2 = sum_i a_i^2 B^(2i) + sum_i a_i B^i * (2 * A_{>i}),   A_{>i} = sum_{j>i} a_j B^j.
    Words of 2*A_{>i}: E_{i+1} = a_{i+1} << 1 at position i+1 (no carry-in: a_i is not part of A_{>i}),
    D_j = (a_j << 1) | (a_{j-1} >> 31) for j > i+1 (a < 2^255, so nothing spills into word 8).
    Operands: %0-7 r, %8-15 a, %16-18 P1..P3.  Scratch: v2-v13 as in the product, v14-v19 = D_2..D_7, v20 = E."""
    a_ = lambda i: f"%{8 + i}"
    t_ = lambda i: f"%{i}"
    Pq = {1: "%16", 2: "%17", 3: "%18"}
    D = {j: f"v{12 + j}" for j in range(2, 8)}       # D_2..D_7 -> v14..v19
    E = "v20"
    L = []
    X, Y = A, B
    have_D = set()
    for k in range(15):
        prods = []           # (x, y, pre) pre = instruction emitted before the MAC
        for i in range(8):
            j = k - i
            if not (0 <= j < 8) or i > j:
                continue
            if i == j:
                prods.append((a_(i), a_(i), None))
            elif j == i + 1:
                prods.append((a_(i), E, f"v_lshlrev_b32 {E}, 1, {a_(j)}"))
            else:
                pre = None
                if j not in have_D:
                    pre = f"v_alignbit_b32 {D[j]}, {a_(j)}, {a_(j - 1)}, 31"
                    have_D.add(j)
                prods.append((a_(i), D[j], pre))
        for pj in (1, 2, 3):
            if 0 <= k - pj < 8:
                prods.append((M[k - pj], Pq[pj], None))
        if 0 <= k - 7 < 8:
            prods.append((M[k - 7], "2.0", None))
        first = True
        for (x, y, pre) in prods:
            if pre:
                L.append(pre)
            if k == 0 and first:
                L.append(f"v_mad_u64_u32 {X[0]}, vcc, {x}, {y}, 0")
            else:
                L.append(f"v_mad_u64_u32 {X[0]}, vcc, {x}, {y}, {X[0]}")
            L.append(f"v_addc_co_u32 {Y[2]}, vcc, 0, {'0' if first else Y[2]}, vcc")
            first = False
        if k < 8:
            L.append(f"v_sub_u32 {M[k]}, 0, {X[1]}")
            L.append(f"v_mad_u64_u32 {X[0]}, vcc, {M[k]}, 1, {X[0]}")
            L.append(f"v_addc_co_u32 {Y[2]}, vcc, 0, {Y[2]}, vcc")
        else:
            L.append(f"v_mov_b32 {t_(k - 8)}, {X[1]}")
        if k < 14:
            L.append(f"v_mov_b32 {Y[1]}, {X[2]}")
            X, Y = Y, X
        else:
            L.append(f"v_mov_b32 {t_(7)}, {X[2]}")
    L.append(f"v_subrev_co_u32 {M[0]}, vcc, 1, {t_(0)}")
    for i in (1, 2, 3):
        L.append(f"v_subb_co_u32 {M[i]}, vcc, {t_(i)}, {Pq[i]}, vcc")
    for i in (4, 5, 6):
        L.append(f"v_subbrev_co_u32 {M[i]}, vcc, 0, {t_(i)}, vcc")
    L.append(f"v_subbrev_co_u32 {M[7]}, vcc, 2.0, {t_(7)}, vcc")
    for i in range(8):
        L.append(f"v_cndmask_b32 {t_(i)}, {M[i]}, {t_(i)}, vcc")
    return L


def gen_sub():
    """r = a - b mod p.  %0-7 r (early clobber), %8-15 a, %16-23 b, %24-26 P1..P3.  Scratch v2-v7."""
    r = lambda i: f"%{i}"; a_ = lambda i: f"%{8 + i}"; b_ = lambda i: f"%{16 + i}"
    L = [f"v_sub_co_u32 {r(0)}, vcc, {a_(0)}, {b_(0)}"]
    for i in range(1, 8):
        L.append(f"v_subb_co_u32 {r(i)}, vcc, {a_(i)}, {b_(i)}, vcc")
    L.append("v_cndmask_b32_e64 v2, 0, -1, vcc")              # all-ones if we borrowed: add p back
    L.append("v_and_b32 v3, 1, v2")
    L.append("v_and_b32 v4, %24, v2")
    L.append("v_and_b32 v5, %25, v2")
    L.append("v_and_b32 v6, %26, v2")
    L.append("v_and_b32 v7, 2.0, v2")
    L.append(f"v_add_co_u32 {r(0)}, vcc, {r(0)}, v3")
    L.append(f"v_addc_co_u32 {r(1)}, vcc, {r(1)}, v4, vcc")
    L.append(f"v_addc_co_u32 {r(2)}, vcc, {r(2)}, v5, vcc")
    L.append(f"v_addc_co_u32 {r(3)}, vcc, {r(3)}, v6, vcc")
    for i in (4, 5, 6):
        L.append(f"v_addc_co_u32 {r(i)}, vcc, 0, {r(i)}, vcc")
    L.append(f"v_addc_co_u32 {r(7)}, vcc, {r(7)}, v7, vcc")
    return L


def gen_add():
    """r = a + b mod p (a + b < 2p < 2^256).  Same operand map as gen_sub.  Scratch v2-v9."""
    r = lambda i: f"%{i}"; a_ = lambda i: f"%{8 + i}"; b_ = lambda i: f"%{16 + i}"
    S = [f"v{2 + i}" for i in range(8)]
    L = [f"v_add_co_u32 {r(0)}, vcc, {a_(0)}, {b_(0)}"]
    for i in range(1, 8):
        L.append(f"v_addc_co_u32 {r(i)}, vcc, {a_(i)}, {b_(i)}, vcc")
    L.append(f"v_subrev_co_u32 {S[0]}, vcc, 1, {r(0)}")
    for i in (1, 2, 3):
        L.append(f"v_subb_co_u32 {S[i]}, vcc, {r(i)}, %{23 + i}, vcc")
    for i in (4, 5, 6):
        L.append(f"v_subbrev_co_u32 {S[i]}, vcc, 0, {r(i)}, vcc")
    L.append(f"v_subbrev_co_u32 {S[7]}, vcc, 2.0, {r(7)}, vcc")
    for i in range(8):
        L.append(f"v_cndmask_b32 {r(i)}, {S[i]}, {r(i)}, vcc")
    return L


out = ["// GENERATED by tools/gen_field_asm.py -- do not edit.  See that file for the schedule.",
       f"// {len(gen_mul())} instructions per Montgomery product.",
       emit("KH_MONT_MUL_ASM", gen_mul()),
       '#define KH_MONT_MUL_CLOBBERS "vcc", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13"',
       f"// Montgomery squaring: {len(gen_sqr())} instructions (36 limb products instead of 64)",
       emit("KH_MONT_SQR_ASM", gen_sqr()),
       '#define KH_MONT_SQR_CLOBBERS "vcc", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20"',
       f"// modular subtraction ({len(gen_sub())} instructions) and addition ({len(gen_add())} instructions)",
       emit("KH_FE_SUB_ASM", gen_sub()),
       '#define KH_FE_SUB_CLOBBERS "vcc", "v2", "v3", "v4", "v5", "v6", "v7"',
       emit("KH_FE_ADD_ASM", gen_add()),
       '#define KH_FE_ADD_CLOBBERS "vcc", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9"',
       ""]
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "proof_systems_amd", "csrc", "field_mulasm.inc")
open(path, "w").write("\n".join(out))
print(len(gen_mul()), "instructions")
