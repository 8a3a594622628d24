"""Limb-exact model of csrc/ntt29.cuh (the NTT butterflies on nine 29-bit limbs): the helpers -- Spread29 / sub29, reduce29, the load and store
conversions -- and one whole radix-4 step + the lone radix-2 stage are replayed on Python integers with every precondition asserted (no limb
negative or >= 2^32, the operand-size bound of the product, the magnitude invariant "normalised and < 2.1 p at step entry"), at random and at
the extreme values the invariant allows, both fields; results are compared with plain modular arithmetic.  The product itself is the
generated asm block that tests/test_field29_model.py / tools/gen_field29_asm.py --check already hold to big integers: here it is modelled by its
contract, (a b + m p) / 2^261 with limbs normalised."""
import random

import pytest

MASK = (1 << 29) - 1
PS = {"fp": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001, "fq": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}
RP = 1 << 261


def limbs(x, norm=True):
    return [(x >> (29 * i)) & MASK for i in range(8)] + [x >> 232]


def value(l):
    return sum(v << (29 * i) for i, v in enumerate(l))


def u32(x):
    assert 0 <= x < (1 << 32), hex(x)
    return x


def spread(p, K, J):
    c = limbs(K * p)
    s = [c[k] + J * (1 << 29) - (J if k > 0 else 0) for k in range(8)] + [c[8] - J]
    assert value(s) == K * p and all(0 <= v < (1 << 32) for v in s)
    return s


def sub29(p, a, b, K, J):
    S = spread(p, K, J)
    assert all(a[i] < (1 << 30) for i in range(9)), "a limb too large for the 32-bit sum"
    assert all(b[i] <= J * MASK for i in range(8)) and b[8] <= S[8], "Spread29 precondition"
    r, carry = [], 0
    for i in range(9):
        t = u32(a[i] + S[i] + carry) - b[i]
        u32(t)
        if i < 8:
            r.append(t & MASK); carry = t >> 29
        else:
            r.append(t)
    assert value(r) == value(a) - value(b) + K * p
    return r


def reduce29(p, x):
    assert all(v < (1 << 32) - 8 for v in x) and value(x) < 64 * p
    pl = limbs(p)
    r, c = [], 0
    for i in range(8):
        t = u32(x[i] + c); r.append(t & MASK); c = t >> 29
    top = u32(x[8] + c)
    u = (top >> 22) - 1
    carry = 0
    for i in range(5):
        t = r[i] - u * pl[i] + carry
        assert -(1 << 63) <= t < (1 << 63)
        r[i] = t & MASK; carry = t >> 29
    cs = carry
    assert -(1 << 31) <= cs < (1 << 31)
    for i in range(5, 8):
        t = r[i] + cs
        assert -(1 << 31) <= t < (1 << 31)
        r[i] = t & MASK; cs = t >> 29
    last = (top & ((1 << 22) - 1)) + (1 << 22) + cs
    r.append(u32(last))
    v = value(r)
    assert v == value(x) - u * p and 0 < v < (1 << 254) + p and all(l <= MASK for l in r[:8])
    return r


def mul29(p, a, b):
    assert all(v < (1 << 32) for v in a + b)
    A = max(max(a).bit_length(), 1); B = max(max(b).bit_length(), 1)
    assert A + B <= 60, (A, B)                                  # 9 2^(A+B) + 4 2^58 + ... < 2^64 (tools/gen_field29_asm.py)
    ab = value(a) * value(b)
    m = (-ab * pow(p, -1, RP)) % RP
    r = (ab + m * p) // RP
    assert r < ab // RP + p + 1
    return limbs(r)


def pack(X, sh):
    return limbs(X << sh)


def wire_times(p, v, W):
    t = mul29(p, v, pack(W, 0))
    x = value(t)
    assert x < 2 * p and x < (1 << 256)
    return x - p if x >= p else x


@pytest.mark.parametrize("field", ["fp", "fq"])
def test_step_helpers_at_random_and_extreme_values(field):
    p = PS[field]
    R, Rp = pow(2, 256, p), pow(2, 261, p)
    rnd = random.Random(29)
    lazy = lambda x: x * Rp % p                                  # the residue a lazy value must be congruent to
    one_lazy = reduce29(p, pack(R, 5))                           # stage twiddle 1: wire one through the load path
    assert value(one_lazy) % p == Rp
    edge = [0, 1, p - 1, p, 2 * p - 1, int(2.1 * p) - 1, (1 << 254) + p - 1, (1 << 254) - 1, 1 << 254]
    for trial in range(400):
        # ---- load: wire -> lazy
        X = rnd.choice([0, 1, p - 1, rnd.randrange(p)]) if trial < 40 else rnd.randrange(p)
        x = reduce29(p, pack(X, 5))
        assert value(x) % p == 32 * X % p and value(x) < 2 * p
        # ---- a radix-4 step on four values at the invariant's edge (normalised, < 2.1 p) or random
        vals = [rnd.choice(edge) if trial < 200 else rnd.randrange(int(2.1 * p)) for _ in range(4)]
        xs = [limbs(v) for v in vals]
        tw = [reduce29(p, pack(rnd.randrange(p), 5)) for _ in range(3)]   # stage twiddles as the kernel prepares them (< 2 p)
        if trial % 5 == 0:
            tw[0] = one_lazy
        add = lambda a, b: [u32(a[i] + b[i]) for i in range(9)]
        s0, s1 = add(xs[0], xs[2]), add(xs[1], xs[3])
        d0 = mul29(p, sub29(p, xs[0], xs[2], 4, 1), tw[0])
        d1 = mul29(p, sub29(p, xs[1], xs[3], 4, 1), tw[1])
        y0 = reduce29(p, add(s0, s1))
        y1 = mul29(p, sub29(p, s0, s1, 6, 2), tw[2])
        y2 = reduce29(p, add(d0, d1))
        y3 = mul29(p, sub29(p, d0, d1, 4, 1), tw[2])
        inv = pow(Rp, -1, p)
        w = [value(t) * inv % p for t in tw]                     # the field elements the twiddles stand for
        f = [v * inv % p for v in vals]
        want = [(f[0] + f[2] + f[1] + f[3]) % p, (f[0] + f[2] - f[1] - f[3]) * w[2] % p,
                ((f[0] - f[2]) * w[0] + (f[1] - f[3]) * w[1]) % p, ((f[0] - f[2]) * w[0] - (f[1] - f[3]) * w[1]) * w[2] % p]
        for y, wv in zip((y0, y1, y2, y3), want):
            assert value(y) % p == lazy(wv) and value(y) < int(2.1 * p) and all(l <= MASK for l in y[:8]), "step output breaks the invariant"
        # ---- the last step of a pass skips its three unit twiddles: outputs go to the store un-multiplied and un-reduced
        dd0 = sub29(p, xs[0], xs[2], 4, 1)
        z = [add(s0, s1), sub29(p, s0, s1, 6, 2), add(dd0, d1), sub29(p, dd0, d1, 4, 1)]
        wantz = [want[0], (f[0] + f[2] - f[1] - f[3]) % p, ((f[0] - f[2]) + (f[1] - f[3]) * w[1]) % p, ((f[0] - f[2]) - (f[1] - f[3]) * w[1]) % p]
        W = rnd.randrange(p)                                     # the inter-pass twiddle / 1/N / one, in wire form
        for y, wv in zip(z, wantz):
            assert value(y) < 11 * p
            assert wire_times(p, y, W) == wv * (W * pow(R, -1, p) % p) * R % p
        # ---- the lone radix-2 stage of an odd pass
        s = reduce29(p, add(xs[0], xs[1])); d = mul29(p, sub29(p, xs[0], xs[1], 4, 1), tw[0])
        assert value(s) % p == lazy((f[0] + f[1]) % p) and value(d) % p == lazy((f[0] - f[1]) * w[0] % p) and value(d) < 1.1 * p
        # ---- coset twiddle on load (first pass of an extension): lazy < 2 p times pack29<5>(wire) < 32 p
        Wc = rnd.randrange(p)
        xc = mul29(p, x, pack(Wc, 5))
        assert value(xc) % p == lazy((X * pow(R, -1, p)) * (Wc * pow(R, -1, p)) % p) and value(xc) < 1.6 * p
