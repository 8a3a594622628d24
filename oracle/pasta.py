"""CPU oracle (Python big-int) for the Kimchi MSM + NTT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``proof_systems_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

It restates, with textbook arithmetic, what the reference computes on the hot
path.  The reference's arithmetic lives in un-vendored crates (ark-ff / ark-ec /
ark-poly 0.5.0, Cargo.lock:171-281); MSM and DFT results are unique
mathematical objects, so parity is defined on results, and this file is pinned
against every golden vector the reference holds for the path (see
``tests/test_oracle_kats.py``):

* constants ............ curves/src/pasta/fields/fp.rs:8-80, fq.rs:8-79,
                         curves/src/pasta/curves/{vesta,pallas}.rs
* SRS::create .......... poly-commitment/src/ipa.rs:751-778, 234-265,
                         groupmap/src/lib.rs:74-189
* point codec .......... utils/src/serialization.rs:67-104 (ark-serialize
                         compressed SW), poly-commitment/src/precomputed_srs.rs:76-91
* test RNG ............. utils/src/lib.rs:83-91 (rand 0.8.5 StdRng = ChaCha12)
* commit / chunk / mask  poly-commitment/src/ipa.rs:605-683
* commit_evaluations ... poly-commitment/src/ipa.rs:706-728,
                         poly-commitment/src/commitment.rs:350-394
* lagrange_basis ....... poly-commitment/src/ipa.rs:1065-1172
* domains .............. kimchi/src/circuits/domains.rs:40-69
"""
from __future__ import annotations

import hashlib
import struct
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

# ---------------------------------------------------------------------------
# Fields (curves/src/pasta/fields/fp.rs:8-12, fq.rs:8-12)
# ---------------------------------------------------------------------------
FP_MODULUS = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
FQ_MODULUS = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
TWO_ADICITY = 32
GENERATOR = 5  # multiplicative generator of both fields (fp.rs:9, fq.rs:9)
R_BITS = 256   # Montgomery radix 2^256 (4 x u64 limbs)


@dataclass(frozen=True)
class Field:
    name: str
    p: int

    @property
    def t(self) -> int:            # odd part of p-1
        return (self.p - 1) >> TWO_ADICITY

    @property
    def two_adic_root(self) -> int:  # 5^T, fp.rs:21-26 / fq.rs:16-24
        return pow(GENERATOR, self.t, self.p)

    @property
    def R(self) -> int:
        return (1 << R_BITS) % self.p

    @property
    def R2(self) -> int:
        return (self.R * self.R) % self.p

    @property
    def inv64(self) -> int:        # -p^{-1} mod 2^64
        return (-pow(self.p, -1, 1 << 64)) % (1 << 64)

    def inv(self, a: int) -> int:
        return pow(a, self.p - 2, self.p)

    def is_square(self, a: int) -> bool:
        return a == 0 or pow(a, (self.p - 1) // 2, self.p) == 1

    def sqrt(self, a: int) -> Optional[int]:
        """Tonelli-Shanks exactly as ark-ff's SqrtPrecomputation::TonelliShanks
        (SURVEY Appendix A.2): no sign normalisation of the root."""
        p = self.p
        a %= p
        if a == 0:
            return 0
        if pow(a, (p - 1) // 2, p) != 1:
            return None
        z = self.two_adic_root
        w = pow(a, (self.t - 1) // 2, p)
        x = a * w % p
        b = x * w % p
        v = TWO_ADICITY
        while b != 1:
            k = 0
            b2k = b
            while b2k != 1:
                b2k = b2k * b2k % p
                k += 1
            w = pow(z, 1 << (v - k - 1), p)
            z = w * w % p
            b = b * z % p
            x = x * w % p
            v = k
        return x

    def to_mont(self, a: int) -> int:
        return (a << R_BITS) % self.p

    def from_mont(self, a: int) -> int:
        return a * pow(1 << R_BITS, -1, self.p) % self.p

    def root_of_unity(self, log2_n: int) -> int:
        """omega_{2^k} = (5^T)^(2^(32-k)) -- kimchi/src/circuits/domains.rs:40-69,
        ark-poly Radix2EvaluationDomain::new."""
        assert 0 <= log2_n <= TWO_ADICITY
        return pow(self.two_adic_root, 1 << (TWO_ADICITY - log2_n), self.p)


Fp = Field("Fp", FP_MODULUS)
Fq = Field("Fq", FQ_MODULUS)

Affine = Optional[Tuple[int, int]]  # None = point at infinity


# ---------------------------------------------------------------------------
# Curves (curves/src/pasta/curves/vesta.rs:8-43, pallas.rs:8-41): y^2 = x^3 + 5
# ---------------------------------------------------------------------------
@dataclass(frozen=True)
class Curve:
    name: str
    cid: int          # C-ABI curve id: 0 = Vesta, 1 = Pallas
    base: Field       # coordinate field
    scalar: Field     # scalar field
    gen: Tuple[int, int]
    b: int = 5

    # -- affine (textbook) ---------------------------------------------------
    def is_on_curve(self, P: Affine) -> bool:
        if P is None:
            return True
        x, y = P
        p = self.base.p
        return (y * y - x * x * x - self.b) % p == 0

    def neg(self, P: Affine) -> Affine:
        if P is None:
            return None
        return (P[0], (-P[1]) % self.base.p)

    def add(self, P: Affine, Q: Affine) -> Affine:
        p = self.base.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, p - 2, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, p - 2, p) % p
        x3 = (lam * lam - x1 - x2) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return (x3, y3)

    # -- Jacobian (fast path for the oracle's own MSMs) ------------------------
    def _jdbl(self, P):
        X, Y, Z = P
        p = self.base.p
        if Z == 0 or Y == 0:
            return (1, 1, 0)
        A = X * X % p
        B = Y * Y % p
        C = B * B % p
        D = 2 * ((X + B) * (X + B) - A - C) % p
        E = 3 * A % p
        F = E * E % p
        X3 = (F - 2 * D) % p
        Y3 = (E * (D - X3) - 8 * C) % p
        Z3 = 2 * Y * Z % p
        return (X3, Y3, Z3)

    def _jadd(self, P, Q):
        p = self.base.p
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        if Z1 == 0:
            return Q
        if Z2 == 0:
            return P
        Z1Z1 = Z1 * Z1 % p
        Z2Z2 = Z2 * Z2 % p
        U1 = X1 * Z2Z2 % p
        U2 = X2 * Z1Z1 % p
        S1 = Y1 * Z2 * Z2Z2 % p
        S2 = Y2 * Z1 * Z1Z1 % p
        if U1 == U2:
            if S1 == S2:
                return self._jdbl(P)
            return (1, 1, 0)
        H = (U2 - U1) % p
        Rr = (S2 - S1) % p
        HH = H * H % p
        HHH = H * HH % p
        V = U1 * HH % p
        X3 = (Rr * Rr - HHH - 2 * V) % p
        Y3 = (Rr * (V - X3) - S1 * HHH) % p
        Z3 = Z1 * Z2 * H % p
        return (X3, Y3, Z3)

    def _to_j(self, P: Affine):
        return (1, 1, 0) if P is None else (P[0], P[1], 1)

    def _from_j(self, P) -> Affine:
        X, Y, Z = P
        if Z == 0:
            return None
        p = self.base.p
        zi = pow(Z, p - 2, p)
        zi2 = zi * zi % p
        return (X * zi2 % p, Y * zi2 * zi % p)

    def mul(self, P: Affine, k: int) -> Affine:
        k %= self.scalar.p
        acc = (1, 1, 0)
        base = self._to_j(P)
        while k:
            if k & 1:
                acc = self._jadd(acc, base)
            base = self._jdbl(base)
            k >>= 1
        return self._from_j(acc)

    def msm(self, points: Sequence[Affine], scalars: Sequence[int]) -> Affine:
        """Sum_i scalars[i] * points[i]; uses min(len) pairs like ark-ec's
        msm_bigint (poly-commitment/src/ipa.rs:663-676 relies on that).
        Simple windowed bucket method; result is the unique group element."""
        n = min(len(points), len(scalars))
        if n == 0:
            return None
        q = self.scalar.p
        sc = [s % q for s in scalars[:n]]
        c = 4 if n < 32 else min(12, max(4, n.bit_length() - 2))
        nwin = (255 + c - 1) // c
        total = (1, 1, 0)
        jp = [self._to_j(P) for P in points[:n]]
        for w in reversed(range(nwin)):
            for _ in range(c):
                total = self._jdbl(total)
            buckets = [(1, 1, 0)] * ((1 << c) - 1)
            shift = w * c
            mask = (1 << c) - 1
            for P, s in zip(jp, sc):
                d = (s >> shift) & mask
                if d:
                    buckets[d - 1] = self._jadd(buckets[d - 1], P)
            run = (1, 1, 0)
            acc = (1, 1, 0)
            for bkt in reversed(buckets):
                run = self._jadd(run, bkt)
                acc = self._jadd(acc, run)
            total = self._jadd(total, acc)
        return self._from_j(total)

    def msm_naive(self, points: Sequence[Affine], scalars: Sequence[int]) -> Affine:
        acc: Affine = None
        for P, s in zip(points, scalars):
            acc = self.add(acc, self.mul(P, s))
        return acc

    # -- SvdW group map + SRS generator --------------------------------------
    def _bw_params(self):
        F = self.base
        p = F.p
        u = 1
        while (u * u * u + self.b) % p == 0:   # groupmap/src/lib.rs:137-145
            u += 1
        fu = (u * u * u + self.b) % p
        three_u2 = 3 * u * u % p
        inv_three_u2 = F.inv(three_u2)
        s = F.sqrt((-three_u2) % p)
        assert s is not None
        c1 = (s - u) * F.inv(2) % p
        return u, fu, c1, s, inv_three_u2

    def to_group(self, t: int) -> Tuple[int, int]:
        """groupmap/src/lib.rs:74-133,184-188 (BWParameters::to_group)."""
        F = self.base
        p = F.p
        u, fu, c1, s, c2 = self._bw_params()
        t2 = t * t % p
        alpha_inv = (t2 + fu) * t2 % p
        alpha = F.inv(alpha_inv) if alpha_inv else 0
        x1 = (c1 - t2 * t2 % p * alpha % p * s) % p
        x2 = (-u - x1) % p
        tpf = (t2 + fu) % p
        x3 = (u - tpf * tpf % p * (alpha * tpf % p) % p * c2) % p
        for x in (x1, x2, x3):
            y = F.sqrt((x * x * x + self.b) % p)
            if y is not None:
                return (x, y)
        raise AssertionError("get_xy")

    def point_of_random_bytes(self, rb: bytes) -> Tuple[int, int]:
        """poly-commitment/src/ipa.rs:234-265: 31 bytes -> 248 bits, LSB-first
        within each byte, consumed big-endian."""
        t = 0
        for i in range(31):
            for j in range(8):
                t = (t << 1) | ((rb[i] >> j) & 1)
        return self.to_group(t)

    def srs_g(self, i: int) -> Tuple[int, int]:
        """poly-commitment/src/ipa.rs:754-762."""
        return self.point_of_random_bytes(hashlib.blake2b(struct.pack(">I", i), digest_size=64).digest())

    def srs_h(self) -> Tuple[int, int]:
        """poly-commitment/src/ipa.rs:765-772."""
        return self.point_of_random_bytes(
            hashlib.blake2b(b"srs_misc" + struct.pack(">I", 0), digest_size=64).digest())

    def srs_create(self, depth: int) -> List[Tuple[int, int]]:
        return [self.srs_g(i) for i in range(depth)]

    # -- ark-serialize compressed codec --------------------------
    def compress(self, P: Affine) -> bytes:
        if P is None:
            return bytes(32) + b"\x40"
        x, y = P
        flag = 0x80 if y > (self.base.p - 1) // 2 else 0
        return x.to_bytes(32, "little") + bytes([flag])

    def decompress(self, b: bytes) -> Affine:
        assert len(b) == 33
        flag = b[32]
        if flag & 0x40:
            return None
        p = self.base.p
        x = int.from_bytes(b[:32], "little")
        y = self.base.sqrt((x * x * x + self.b) % p)
        assert y is not None, "not on curve"
        neg = y > (p - 1) // 2
        if neg != bool(flag & 0x80):
            y = p - y
        return (x, y)


VESTA = Curve("vesta", 0, Fq, Fp,
              (1, 11426906929455361843568202299992114520848200991084027513389447476559454104162))
PALLAS = Curve("pallas", 1, Fp, Fq,
               (1, 12418654782883325593414442427049395787963493412651469444558597405572177144507))
CURVES = {0: VESTA, 1: PALLAS, "vesta": VESTA, "pallas": PALLAS}


# ---------------------------------------------------------------------------
# Endomorphism constants and scalar challenges (poly-commitment/src/ipa.rs:214-231,
# poseidon/src/sponge.rs:110-114, 190-226)
# ---------------------------------------------------------------------------
def endo_coefficient(F: Field) -> int:
    """GENERATOR^((p-1)/3): a primitive cube root of unity (sponge.rs:110-114)."""
    return pow(GENERATOR, (F.p - 1) // 3, F.p)


def endos(curve: Curve) -> Tuple[int, int]:
    """(endo_q in the base field, endo_r in the scalar field) with phi(P) = (endo_q x, y) = [endo_r] P
    (ipa.rs:214-231)."""
    eq = endo_coefficient(curve.base)
    er = endo_coefficient(curve.scalar)
    g = curve.gen
    phi = (g[0] * eq % curve.base.p, g[1])
    if curve.mul(g, er) != phi:
        er = er * er % curve.scalar.p
        assert curve.mul(g, er) == phi
    return eq, er


def challenge_to_field(F: Field, chal: int, endo: int, length_in_bits: int = 128) -> int:
    """ScalarChallenge::to_field_with_length (sponge.rs:190-215): k = a * endo + b."""
    a = b = 2
    for i in reversed(range(length_in_bits // 2)):
        a = 2 * a % F.p
        b = 2 * b % F.p
        s = 1 if (chal >> (2 * i)) & 1 else F.p - 1
        if (chal >> (2 * i + 1)) & 1 == 0:
            b = (b + s) % F.p
        else:
            a = (a + s) % F.p
    return (a * endo + b) % F.p


def combine_one_endo(curve: Curve, g1: Sequence[Affine], g2: Sequence[Affine], chal: int) -> List[Affine]:
    """CommitmentCurve::combine_one_endo (commitment.rs:581-589 -> combine.rs:292-340): the Halo endo ladder,
    acc = 2 (phi(g2) + g2); for each 2-bit chunk: s = +-g2 [phi'd]; acc = (acc + s) + acc; result g1 + acc."""
    eq, _ = endos(curve)
    p = curve.base.p
    out = []
    for P1, P2 in zip(g1, g2):
        phi = (P2[0] * eq % p, P2[1])
        acc = curve.add(curve.add(phi, P2), curve.add(phi, P2))
        for i in reversed(range(64)):
            S = P2 if (chal >> (2 * i)) & 1 else curve.neg(P2)
            if (chal >> (2 * i + 1)) & 1:
                S = (S[0] * eq % p, S[1])
            acc = curve.add(curve.add(acc, S), acc)
        out.append(curve.add(P1, acc))
    return out


def b_poly_coefficients(F: Field, chals: Sequence[int]) -> List[int]:
    """commitment.rs:464-476, literally: s[i] = s[i - 2^(k-1)] * chals[rounds - k], k = position of i's top bit + 1."""
    rounds = len(chals)
    s = [1] * (1 << rounds)
    k, pw = 0, 1
    for i in range(1, 1 << rounds):
        if i == pw:
            k += 1
            pw <<= 1
        s[i] = s[i - (pw >> 1)] * chals[rounds - 1 - (k - 1)] % F.p
    return s


def ipa_open_rounds(curve: Curve, g: Sequence[Affine], h: Affine, u_base: Affine, a: Sequence[int], b: Sequence[int],
                    rands: Sequence[Tuple[int, int]], chals: Sequence[int], msm=None):
    """The folding loop of SRS::open, literally (ipa.rs:929-1018): per round L = <a_hi, g_lo> + rand_l H + <a_hi, b_lo> U,
    R = <a_lo, g_hi> + rand_r H + <a_lo, b_hi> U, u = u_pre.to_field(endo_r), a = a_lo + u^-1 a_hi, b = b_lo + u b_hi,
    g = combine_one_endo(g_lo, g_hi, u_pre).  The sponge is the caller's: the prechallenges are inputs.
    Returns (lr, us, a0, b0, g0).  `msm(points, scalars)` may be supplied to speed up the L/R sums."""
    F = curve.scalar
    _, endo_r = endos(curve)
    if msm is None:
        def msm(pts, sc):
            acc = None
            for pt, k in zip(pts, sc):
                acc = curve.add(acc, curve.mul(pt, k))
            return acc
    g = list(g); a = list(a) + [0] * (len(g) - len(a)); b = list(b)
    assert len(g) == len(b) and len(g) & (len(g) - 1) == 0
    lr, us = [], []
    for (rand_l, rand_r), u_pre in zip(rands, chals):
        n = len(g) // 2
        g_lo, g_hi, a_lo, a_hi, b_lo, b_hi = g[:n], g[n:], a[:n], a[n:], b[:n], b[n:]
        ip_l = sum(x * y for x, y in zip(a_hi, b_lo)) % F.p
        ip_r = sum(x * y for x, y in zip(a_lo, b_hi)) % F.p
        L = msm(g_lo + [h, u_base], a_hi + [rand_l, ip_l])
        R = msm(g_hi + [h, u_base], a_lo + [rand_r, ip_r])
        lr.append((L, R))
        u = challenge_to_field(F, u_pre, endo_r)
        u_inv = F.inv(u)
        us.append(u)
        a = [(lo + u_inv * hi) % F.p for lo, hi in zip(a_lo, a_hi)]
        b = [(lo + u * hi) % F.p for lo, hi in zip(b_lo, b_hi)]
        g = combine_one_endo(curve, g_lo, g_hi, u_pre)
    assert len(g) == 1
    return lr, us, a[0], b[0], g[0]


# ---------------------------------------------------------------------------
# msgpack SRS file (precomputed_srs.rs:76-91)
# ---------------------------------------------------------------------------
def read_srs_file(path: str, curve: Curve, limit: Optional[int] = None):
    """Returns (list of compressed 33-byte g_i, compressed h)."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[0] == 0x92 and data[1] == 0xDD
    n = struct.unpack(">I", data[2:6])[0]
    off = 6
    pts = []
    for _ in range(n):
        assert data[off] == 0xC4 and data[off + 1] == 0x21
        pts.append(data[off + 2: off + 35])
        off += 35
    assert data[off] == 0xC4 and data[off + 1] == 0x21
    h = data[off + 2: off + 35]
    assert off + 35 == len(data)
    if limit is not None:
        pts = pts[:limit]
    return pts, h


def msgpack_polycomm(curve: Curve, chunks: Sequence[Affine]) -> bytes:
    """PolyComm{chunks} as rmp-serde writes it (tests/commitment.rs:369-383):
    array(1)[ array(n)[ bin8(33) ... ] ]."""
    n = len(chunks)
    assert n < 16
    out = bytes([0x91, 0x90 | n])
    for P in chunks:
        out += b"\xc4\x21" + curve.compress(P)
    return out


# ---------------------------------------------------------------------------
# rand 0.8.5 StdRng (ChaCha12) + ark-ff 0.5 Fp::rand
# ---------------------------------------------------------------------------
class StdRng:
    def __init__(self, seed: bytes):
        assert len(seed) == 32
        self.key = list(struct.unpack("<8I", seed))
        self.counter = 0
        self.buf: List[int] = []

    @staticmethod
    def _rotl(x, n):
        return ((x << n) | (x >> (32 - n))) & 0xFFFFFFFF

    def _block(self):
        c = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
        st = c + self.key + [self.counter & 0xFFFFFFFF, (self.counter >> 32) & 0xFFFFFFFF, 0, 0]
        x = st[:]

        def qr(a, b, c_, d):
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = self._rotl(x[d] ^ x[a], 16)
            x[c_] = (x[c_] + x[d]) & 0xFFFFFFFF; x[b] = self._rotl(x[b] ^ x[c_], 12)
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = self._rotl(x[d] ^ x[a], 8)
            x[c_] = (x[c_] + x[d]) & 0xFFFFFFFF; x[b] = self._rotl(x[b] ^ x[c_], 7)

        for _ in range(6):  # 12 rounds
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        self.counter += 1
        return [(a + b) & 0xFFFFFFFF for a, b in zip(x, st)]

    def next_u32(self) -> int:
        if not self.buf:
            self.buf = self._block()
        return self.buf.pop(0)

    def next_u64(self) -> int:
        lo = self.next_u32()
        hi = self.next_u32()
        return lo | (hi << 32)


def field_rand(F: Field, rng: StdRng) -> int:
    """ark-ff 0.5 `Fp::rand`: 4 u64 limbs, clear the top bit, accept if < p,
    the limbs ARE the Montgomery representation.  Returns the canonical value."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << 63) - 1
        v = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
        if v < F.p:
            return F.from_mont(v)


# ---------------------------------------------------------------------------
# Commitment semantics (poly-commitment/src/ipa.rs:605-748)
# ---------------------------------------------------------------------------
def commit_non_hiding(curve: Curve, g: Sequence[Affine], coeffs: Sequence[int],
                      num_chunks: int) -> List[Affine]:
    """ipa.rs:638-683."""
    q = curve.scalar.p
    coeffs = [c % q for c in coeffs]
    while coeffs and coeffs[-1] == 0:     # DensePolynomial is truncated
        coeffs.pop()
    n = len(g)
    if not coeffs:
        chunks: List[Affine] = [None]
    else:
        chunks = [curve.msm(g, coeffs[i:i + n]) for i in range(0, len(coeffs), n)]
    while len(chunks) < num_chunks:
        chunks.append(None)
    return chunks


def mask_custom(curve: Curve, h: Affine, com: Sequence[Affine], blinders: Sequence[int]) -> List[Affine]:
    """ipa.rs:605-622: C_j + w_j * h."""
    assert len(com) == len(blinders)
    return [curve.add(curve.mul(h, w), c) for c, w in zip(com, blinders)]


def lagrange_basis(curve: Curve, g: Sequence[Affine], log2_n: int) -> List[List[Affine]]:
    """ipa.rs:1065-1172 by definition: the inverse DFT over curve points.
    Returns basis[i] = list of chunks.  O(n^2) group ops -- small n only."""
    n = 1 << log2_n
    F = curve.scalar
    w = F.root_of_unity(log2_n)
    winv = F.inv(w)
    ninv = F.inv(n)
    srs_size = len(g)
    num_elems = (n + srs_size - 1) // srs_size
    out: List[List[Affine]] = [[] for _ in range(n)]
    for c in range(num_elems):
        start = c * srs_size
        num_terms = min((c + 1) * srs_size, n) - start
        for i in range(n):
            # L_i commitment chunk c = n^{-1} sum_j w^{-ij} g[j - start]
            sc = [ninv * pow(winv, i * (start + j), F.p) % F.p for j in range(num_terms)]
            out[i].append(curve.msm(g[:num_terms], sc))
    return out


def commit_evaluations_non_hiding(curve: Curve, basis: Sequence[Sequence[Affine]],
                                  evals: Sequence[int], log2_domain: int) -> List[Affine]:
    """ipa.rs:706-728 + commitment.rs:350-394."""
    n = 1 << log2_domain
    assert len(evals) >= n and len(evals) % n == 0
    s = len(evals) // n
    v = [evals[s * i] for i in range(n)]
    nchunks = max(len(b) for b in basis)
    return [curve.msm([b[c] for b in basis if c < len(b)],
                      [x for b, x in zip(basis, v) if c < len(b)]) for c in range(nchunks)]


# ---------------------------------------------------------------------------
# NTT (ark-poly Radix2EvaluationDomain semantics)
# ---------------------------------------------------------------------------
def _bitrev(n: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def ntt(F: Field, a: Sequence[int], log2_n: int, inverse: bool = False) -> List[int]:
    """Natural order in, natural order out.  Forward: out[j] = sum a_i w^{ij}
    (input zero-padded to N).  Inverse: c_i = N^{-1} sum e_j w^{-ij}."""
    n = 1 << log2_n
    p = F.p
    assert len(a) <= n
    x = [v % p for v in a] + [0] * (n - len(a))
    w = F.root_of_unity(log2_n)
    if inverse:
        w = F.inv(w)
    x = [x[_bitrev(i, log2_n)] for i in range(n)]
    m = 1
    while m < n:
        wm = pow(w, n // (2 * m), p)
        for k in range(0, n, 2 * m):
            t = 1
            for j in range(m):
                u = x[k + j]
                v = x[k + j + m] * t % p
                x[k + j] = (u + v) % p
                x[k + j + m] = (u - v) % p
                t = t * wm % p
        m *= 2
    if inverse:
        ninv = F.inv(n)
        x = [v * ninv % p for v in x]
    return x


def dft_naive(F: Field, a: Sequence[int], log2_n: int, inverse: bool = False) -> List[int]:
    n = 1 << log2_n
    p = F.p
    w = F.root_of_unity(log2_n)
    if inverse:
        w = F.inv(w)
    out = []
    for j in range(n):
        wj = pow(w, j, p)
        acc = 0
        t = 1
        for i, v in enumerate(a):
            acc = (acc + v * t) % p
            t = t * wj % p
        out.append(acc)
    if inverse:
        ninv = F.inv(n)
        out = [v * ninv % p for v in out]
    return out


def lde(F: Field, coeffs: Sequence[int], log2_n: int, log2_blowup: int) -> List[int]:
    """DensePolynomial::evaluate_over_domain_by_ref(d8) (constraints.rs:490-495):
    zero-extend n coefficients to n<<b and forward-NTT."""
    return ntt(F, list(coeffs), log2_n + log2_blowup, inverse=False)


# ---------------------------------------------------------------------------
# wire helpers: 4 x u64 little-endian Montgomery limbs (ark-ff in-memory layout;
# kimchi/src/cached_prover_index.rs:502-539)
# ---------------------------------------------------------------------------
def to_limbs(v: int) -> Tuple[int, int, int, int]:
    m = (1 << 64) - 1
    return (v & m, (v >> 64) & m, (v >> 128) & m, (v >> 192) & m)


def from_limbs(l) -> int:
    return int(l[0]) | (int(l[1]) << 64) | (int(l[2]) << 128) | (int(l[3]) << 192)


# ---------------------------------------------------------------------------
# SRS::open end to end (poly-commitment/src/ipa.rs:811-1063) and the opening-proof KAT generator
# (poly-commitment/tests/commitment.rs:119-231, 388-440).  The heavy steps are injectable so that the same
# transcript logic can drive either the oracle's own arithmetic or the device (tests/test_gpu_open_kat.py).
# ---------------------------------------------------------------------------
def shift_scalar(curve: Curve, x: int) -> int:
    """commitment.rs:273-288."""
    F = curve.scalar
    two_pow = pow(2, F.p.bit_length(), F.p)
    if F.p < curve.base.p:
        return (x - (two_pow + 1)) * F.inv(2) % F.p
    return (x - two_pow) % F.p


def combine_polys(curve: Curve, plnms, polyscale: int, srs_length: int):
    """utils.rs:103-206 for polynomials in coefficient form: p = sum polyscale^i * chunk_i, and the combined blinder."""
    F = curve.scalar
    acc: List[int] = []
    combined_comm, scale = 0, 1
    for coeffs, blinders in plnms:
        offset = 0
        for w in blinders:
            seg = coeffs[min(offset, len(coeffs)): min(offset + srs_length, len(coeffs))]
            if len(seg) > len(acc):
                acc += [0] * (len(seg) - len(acc))
            for i, c in enumerate(seg):
                acc[i] = (acc[i] + scale * c) % F.p
            combined_comm = (combined_comm + w * scale) % F.p
            scale = scale * polyscale % F.p
            offset += srs_length
    while acc and acc[-1] == 0:
        acc.pop()
    return acc, combined_comm


def b_init_vector(F: Field, elm: Sequence[int], evalscale: int, n: int) -> List[int]:
    """ipa.rs:863-888: b[j] = sum_i evalscale^i * elm_i^j."""
    b = [0] * n
    scale = 1
    for e in elm:
        t = 1
        for i in range(n):
            b[i] = (b[i] + scale * t) % F.p
            t = t * e % F.p
        scale = scale * evalscale % F.p
    return b


def ipa_open(curve: Curve, g: Sequence[Affine], h: Affine, plnms, elm: Sequence[int], polyscale: int, evalscale: int,
             sponge, rng, rounds_backend=None, vectors_backend=None):
    """SRS::open (ipa.rs:824-1063) for a power-of-two SRS.  `plnms` = [(coefficients, blinder chunks)], `sponge` an
    oracle.poseidon.DefaultFqSponge, `rng` a StdRng.  rounds_backend(a, b, u_base) may return an object with
    round_lr(rand_l, rand_r) -> (L, R), round_fold(u_pre) -> u and finish() -> (a0, b0, g0) (the device loop);
    by default the rounds are the literal ones of ipa_open_rounds.  vectors_backend(plnms, polyscale, elm, evalscale, n)
    may supply (p, b_init) computed elsewhere (the device's combine_polys / b_init); rounds_backend then receives
    whatever third value it returns.  Returns the OpeningProof as a dict."""
    F = curve.scalar
    _, endo_r = endos(curve)
    n = len(g)
    rounds = n.bit_length() - 1
    assert 1 << rounds == n
    p, blinding_factor = combine_polys(curve, plnms, polyscale, n)
    handle = None
    if vectors_backend:
        p, b_init, handle = vectors_backend(plnms, polyscale, elm, evalscale, n)
    else:
        b_init = b_init_vector(F, elm, evalscale, n)
    cip = sum(x * y for x, y in zip(p, b_init)) % F.p
    sponge.absorb_fr([shift_scalar(curve, cip)])
    u_base = curve.to_group(sponge.challenge_fq())
    a = list(p) + [0] * (n - len(p))

    class _Literal:
        def __init__(self):
            self.g, self.a, self.b = list(g), list(a), list(b_init)
        def round_lr(self, rand_l, rand_r):
            m = len(self.g) // 2
            ip_l = sum(x * y for x, y in zip(self.a[m:], self.b[:m])) % F.p
            ip_r = sum(x * y for x, y in zip(self.a[:m], self.b[m:])) % F.p
            L = curve.msm(self.g[:m] + [h, u_base], self.a[m:] + [rand_l, ip_l])
            R = curve.msm(self.g[m:] + [h, u_base], self.a[:m] + [rand_r, ip_r])
            return L, R
        def round_fold(self, u_pre):
            m = len(self.g) // 2
            u = challenge_to_field(F, u_pre, endo_r); ui = F.inv(u)
            self.a = [(lo + ui * hi) % F.p for lo, hi in zip(self.a[:m], self.a[m:])]
            self.b = [(lo + u * hi) % F.p for lo, hi in zip(self.b[:m], self.b[m:])]
            self.g = combine_one_endo(curve, self.g[:m], self.g[m:], u_pre)
            return u
        def finish(self):
            return self.a[0], self.b[0], self.g[0]

    if rounds_backend:
        st = rounds_backend(a, b_init, u_base, handle) if vectors_backend else rounds_backend(a, b_init, u_base)
    else:
        st = _Literal()
    lr, blinders, chals = [], [], []
    for _ in range(rounds):
        rand_l = field_rand(F, rng); rand_r = field_rand(F, rng)
        L, R = st.round_lr(rand_l, rand_r)
        lr.append((L, R)); blinders.append((rand_l, rand_r))
        sponge.absorb_g([L]); sponge.absorb_g([R])
        u_pre = sponge.challenge()
        chals.append(st.round_fold(u_pre))
    a0, b0, g0 = st.finish()
    r_prime = blinding_factor
    for (rl, rr), u in zip(blinders, chals):
        r_prime = (r_prime + rl * F.inv(u) + rr * u) % F.p
    d = field_rand(F, rng); r_delta = field_rand(F, rng)
    delta = curve.add(curve.mul(curve.add(g0, curve.mul(u_base, b0)), d), curve.mul(h, r_delta))
    sponge.absorb_g([delta])
    c = challenge_to_field(F, sponge.challenge(), endo_r)
    return {"lr": lr, "delta": delta, "z1": (a0 * c + d) % F.p, "z2": (r_prime * c + r_delta) % F.p, "sg": g0,
            "chals": chals, "u_base": u_base}


def msgpack_opening_proof(curve: Curve, proof) -> bytes:
    """OpeningProof{lr, delta, z1, z2, sg} (ipa.rs:1175-1191) as rmp-serde writes it: array(5)[array(k)[array(2)[bin L, bin R]...],
    bin delta, bin32 z1, bin32 z2, bin sg]."""
    k = len(proof["lr"])
    assert k < 16
    out = bytes([0x95, 0x90 | k])
    for L, R in proof["lr"]:
        out += b"\x92\xc4\x21" + curve.compress(L) + b"\xc4\x21" + curve.compress(R)
    out += b"\xc4\x21" + curve.compress(proof["delta"])
    out += b"\xc4\x20" + proof["z1"].to_bytes(32, "little") + b"\xc4\x20" + proof["z2"].to_bytes(32, "little")
    out += b"\xc4\x21" + curve.compress(proof["sg"])
    return out


def first_random_opening_proof(curve: Curve, g, h, rng, sponge, commit=None, rounds_backend=None, vectors_backend=None):
    """proofs[0] of generate_random_opening_proof (tests/commitment.rs:119-231): 7 evaluation points, 11 polynomials of
    random length < 500 committed with srs.commit(.., 1, rng), then open().  commit(coeffs) -> chunks may be injected."""
    F = curve.scalar
    n = len(g)
    elm = [field_rand(F, rng) for _ in range(7)]
    plnms, comms = [], []
    for _ in range(11):
        ln = rng.next_u64() % 500
        coeffs = [] if ln == 0 else [field_rand(F, rng) for _ in range(ln + 1)]
        chunks = commit(coeffs) if commit else commit_non_hiding(curve, g, coeffs, 1)
        blinders = [field_rand(F, rng) for _ in chunks]                      # SRS::mask (ipa.rs:628-635)
        comms.append(mask_custom(curve, h, chunks, blinders))
        plnms.append((coeffs, blinders))
    polymask = field_rand(F, rng); evalmask = field_rand(F, rng)
    verifier_sponge = sponge.clone()
    proof = ipa_open(curve, g, h, plnms, elm, polymask, evalmask, sponge, rng, rounds_backend, vectors_backend)
    # what the verifier is handed (tests/commitment.rs:162-227): chunked evaluations at the 7 points
    evaluations = []
    for (coeffs, _w), com in zip(plnms, comms):
        nch = max(1, -(-len(coeffs) // n))
        ev = []
        for pt in elm:
            row = []
            for ci in range(nch):
                acc = 0
                for v in reversed(coeffs[ci * n:(ci + 1) * n]):
                    acc = (acc * pt + v) % F.p
                row.append(acc)
            ev.append(row)
        evaluations.append((com, ev))
    proof["verifier_input"] = {"sponge": verifier_sponge, "evaluation_points": elm, "polyscale": polymask, "evalscale": evalmask,
                               "evaluations": evaluations, "opening": proof,
                               "combined_inner_product": combined_inner_product(F, polymask, evalmask, [e for _, e in evaluations])}
    return proof, comms


# ---------------------------------------------------------------------------
# SRS::verify (poly-commitment/src/ipa.rs:301-502)
# ---------------------------------------------------------------------------
def b_poly(F: Field, chals: Sequence[int], x: int) -> int:
    """commitment.rs:426-436."""
    k = len(chals)
    pw = [x]
    for _ in range(1, k):
        pw.append(pw[-1] * pw[-1] % F.p)
    r = 1
    for i in range(k):
        r = r * (1 + chals[i] * pw[k - 1 - i]) % F.p
    return r


def combined_inner_product(F: Field, polyscale: int, evalscale: int, polys) -> int:
    """commitment.rs:622-657.  polys[t][j][i] = evaluation of chunk i of polynomial t at point j."""
    res, ps = 0, 1
    for evals_tr in polys:
        if not evals_tr[0]:
            continue
        for i in range(len(evals_tr[0])):
            term = 0
            for j in reversed(range(len(evals_tr))):
                term = (term * evalscale + evals_tr[j][i]) % F.p
            res = (res + ps * term) % F.p
            ps = ps * polyscale % F.p
    return res


def ipa_verify_terms(curve: Curve, n: int, h: Affine, batch, rng):
    """The scalars and points SRS::verify (ipa.rs:301-502) feeds to its one MSM, split into the SRS part and the rest:
    returns (g_terms, extra_points, extra_scalars) with g_terms = [(weight, chals)], i.e. the scalar of g_j is
    sum weight * b_poly_coefficients(chals)[j]; H is extra_points[0].  batch items: dicts with sponge,
    evaluation_points, polyscale, evalscale, evaluations [(commitment chunks, evals[point][chunk])], opening,
    combined_inner_product."""
    F = curve.scalar
    _, endo_r = endos(curve)
    rounds = n.bit_length() - 1
    rand_base = field_rand(F, rng); sg_rand_base = field_rand(F, rng)
    rb_i, sg_i = 1, 1
    pts: List[Affine] = [h]
    sc: List[int] = [0]
    g_terms = []
    for it in batch:
        sponge, op = it["sponge"], it["opening"]
        assert len(op["lr"]) == rounds
        sponge.absorb_fr([shift_scalar(curve, it["combined_inner_product"])])
        u_base = curve.to_group(sponge.challenge_fq())
        chal = []
        for L, R in op["lr"]:
            sponge.absorb_g([L]); sponge.absorb_g([R])
            chal.append(challenge_to_field(F, sponge.challenge(), endo_r))
        chal_inv = [F.inv(u) for u in chal]
        sponge.absorb_g([op["delta"]])
        c = challenge_to_field(F, sponge.challenge(), endo_r)
        b0, scale = 0, 1
        for e in it["evaluation_points"]:
            b0 = (b0 + scale * b_poly(F, chal, e)) % F.p
            scale = scale * it["evalscale"] % F.p
        pts.append(op["sg"]); sc.append((-rb_i * op["z1"] - sg_i) % F.p)
        g_terms.append((sg_i, chal))
        sc[0] = (sc[0] - rb_i * op["z2"]) % F.p
        pts.append(u_base); sc.append(-rb_i * (op["z1"] * b0) % F.p)
        rc = c * rb_i % F.p
        for (L, R), ui, u in zip(op["lr"], chal_inv, chal):
            pts.append(L); sc.append(rc * ui % F.p)
            pts.append(R); sc.append(rc * u % F.p)
        ps = 1                                           # combine_commitments (commitment.rs:724-744)
        for chunks, _ev in it["evaluations"]:
            for ch in chunks:
                pts.append(ch); sc.append(rc * ps % F.p)
                ps = ps * it["polyscale"] % F.p
        pts.append(u_base); sc.append(rc * it["combined_inner_product"] % F.p)
        pts.append(op["delta"]); sc.append(rb_i)
        rb_i = rb_i * rand_base % F.p
        sg_i = sg_i * sg_rand_base % F.p
    return g_terms, pts, sc


def ipa_verify(curve: Curve, g: Sequence[Affine], h: Affine, batch, rng) -> bool:
    """SRS::verify with the final MSM done by the oracle."""
    F = curve.scalar
    g_terms, pts, sc = ipa_verify_terms(curve, len(g), h, batch, rng)
    gs = [0] * len(g)
    for w, chal in g_terms:
        for j, s in enumerate(b_poly_coefficients(F, chal)):
            gs[j] = (gs[j] + w * s) % F.p
    return curve.msm(list(g) + pts, gs + sc) is None


# ---------------------------------------------------------------------------
# PolishToken machine per row (kimchi/src/circuits/expr.rs:856-937 with the column addressing of
# Expr::evaluations, expr.rs:1972-1987).  Opcodes as in include/kimchi_hip.h.
# ---------------------------------------------------------------------------
TOK_CONST, TOK_CELL, TOK_DUP, TOK_POW, TOK_ADD, TOK_MUL, TOK_SUB, TOK_STORE, TOK_LOAD = range(9)


def polish_evaluate_rows(F: Field, tokens, cols, consts, rows: int, stride: int = 1, next_shift: int = 8) -> List[int]:
    out = []
    for i in range(rows):
        stack, cache = [], []
        for op, arg in tokens:
            if op == TOK_CONST:
                stack.append(consts[arg])
            elif op == TOK_CELL:
                col = cols[arg >> 1]
                stack.append(col[(stride * i + (next_shift if arg & 1 else 0)) % len(col)])
            elif op == TOK_DUP:
                stack.append(stack[-1])
            elif op == TOK_POW:
                stack[-1] = pow(stack[-1], arg, F.p)
            elif op in (TOK_ADD, TOK_MUL, TOK_SUB):
                y = stack.pop(); x = stack.pop()
                stack.append((x + y) % F.p if op == TOK_ADD else (x * y) % F.p if op == TOK_MUL else (x - y) % F.p)
            elif op == TOK_STORE:
                cache.append(stack[-1])
            elif op == TOK_LOAD:
                stack.append(cache[arg])
            else:
                raise ValueError(op)
        assert len(stack) == 1
        out.append(stack[0])
    return out


def generic_gate_tokens(w0: int, c0: int, sel: int, alpha0: int, alpha1: int):
    """index(Generic) * (alpha^a0 * constraint1 + alpha^a1 * constraint2) of the double generic gate
    (kimchi/src/circuits/polynomials/generic.rs:83-120, argument.rs:201-214) in reverse Polish form.
    w0 = column index of witness 0 (witness columns w0..w0+5), c0 = column index of coefficient 0 (c0..c0+9),
    sel = the selector column, alpha0/alpha1 = indices of the two alpha powers in the constants table."""
    C_ = lambda col: (TOK_CELL, 2 * col)
    t = [C_(sel)]
    for g, alpha in ((0, alpha0), (1, alpha1)):
        w, c = w0 + 3 * g, c0 + 5 * g
        t += [(TOK_CONST, alpha)]
        t += [C_(c), C_(w), (TOK_MUL, 0)]                                  # left_coeff * left
        t += [C_(c + 1), C_(w + 1), (TOK_MUL, 0), (TOK_ADD, 0)]            # + right_coeff * right
        t += [C_(c + 2), C_(w + 2), (TOK_MUL, 0), (TOK_ADD, 0)]            # + out_coeff * out
        t += [C_(c + 3), C_(w), (TOK_MUL, 0), C_(w + 1), (TOK_MUL, 0), (TOK_ADD, 0)]   # + mul_coeff * left * right
        t += [C_(c + 4), (TOK_ADD, 0)]                                     # + constant
        t += [(TOK_MUL, 0)]                                                # alpha^a * constraint
        if g == 1:
            t += [(TOK_ADD, 0)]
    t += [(TOK_MUL, 0)]                                                    # selector * (...)
    return t


# ---------------------------------------------------------------------------
# Permutation argument (kimchi/src/circuits/polynomials/permutation.rs)
# ---------------------------------------------------------------------------
def perm_aggreg(F: Field, witness, sigma, shifts, sid, beta: int, gamma: int, zk_rows: int, rands):
    """perm_aggreg (permutation.rs:447-577), literally: z[0] = 1, z[j+1] = z[j] * prod_i (w_i[j] + sid[j] beta shift_i + gamma)
    / prod_i (w_i[j] + sigma_i[j] beta + gamma), except that z[n - zk_rows + 1] and z[n - zk_rows + 2] are random.
    witness / sigma: PERMUTS columns of n values on d1; rands: the two random values, in draw order."""
    n = len(sid)
    den = [1] * n; num = [1] * n
    for w, s_col, sh in zip(witness, sigma, shifts):
        for j in range(n - 1):
            den[j + 1] = den[j + 1] * (w[j] + s_col[j] * beta + gamma) % F.p
            num[j + 1] = num[j + 1] * (w[j] + sid[j] * beta * sh + gamma) % F.p
    z = [1] + [F.inv(d) if d else 0 for d in den[1:]]
    it = iter(rands)
    for j in range(n - 1):
        if j != n - zk_rows and j != n - zk_rows + 1:
            z[j + 1] = z[j + 1] * num[j + 1] * z[j] % F.p
        else:
            z[j + 1] = next(it)
    return z


def perm_quot_tokens(w0: int, s0: int, z: int, x: int, zkpm: int, gamma: int, beta: int, bshift0: int, alpha0: int, permuts: int = 7):
    """The `perm` part of perm_quot (permutation.rs:237-283) in reverse Polish form:
    alpha0 * zkpm(x) * ( z(x) prod_i (w_i + gamma + x beta shift_i)  -  z(x w) prod_i (w_i + gamma + sigma_i beta) ).
    Column indices: witness w0.., sigma s0.., z, x (poly_x_d1), zkpm (permutation_vanishing_polynomial_l); constants:
    gamma, beta, beta*shift_i at bshift0.., alpha0."""
    C_ = lambda col, nxt=0: (TOK_CELL, 2 * col + nxt)
    t = []
    for i in range(permuts):                                # shifts
        t += [C_(w0 + i), (TOK_CONST, gamma), (TOK_ADD, 0), C_(x), (TOK_CONST, bshift0 + i), (TOK_MUL, 0), (TOK_ADD, 0)]
        if i:
            t += [(TOK_MUL, 0)]
    t += [C_(z), (TOK_MUL, 0)]
    for i in range(permuts):                                # sigmas
        t += [C_(w0 + i), (TOK_CONST, gamma), C_(s0 + i), (TOK_CONST, beta), (TOK_MUL, 0), (TOK_ADD, 0), (TOK_ADD, 0)]
        if i:
            t += [(TOK_MUL, 0)]
    t += [C_(z, 1), (TOK_MUL, 0), (TOK_SUB, 0), (TOK_CONST, alpha0), (TOK_MUL, 0), C_(zkpm), (TOK_MUL, 0)]
    return t
