// field29.cuh -- Pasta field arithmetic on NINE 29-bit limbs with lazy reduction, for the MSM accumulation kernel.
//
// field.cuh keeps elements as eight 32-bit limbs, always fully reduced: every limb product costs a v_mad_u64_u32
// PLUS a v_addc_co_u32 (the 64-bit column accumulator overflows), 254 instructions per Montgomery product.  Here a
// 29 x 29-bit limb product is 58 bits, a whole column of the product scan fits one 64-bit accumulator, and the
// product is 131 MADs + 35 bookkeeping instructions = 166 (squaring 138; tools/gen_field29_asm.py, which also
// checks the generated instruction streams and the limb model of madd29 below against big-integer arithmetic).
//
//   value(a) = sum a.v[i] 2^(29 i);  Montgomery radix R' = 2^261 = 128 p-ish, so there are 6 spare bits:
//   mul29(a, b) = a b / R' mod p  <  a b / R' + p  for ANY a, b with a b < R' p  -- no final subtraction, operands may be
//   several p large.  Additions and subtractions are limb-wise without carry chains (a - b is a + (K p spread over
//   the limbs) - b), followed by ONE carry sweep where the next use needs normalised limbs (< 2^29).
//   Accumulator bound of a product: limbs a_i < 2^A, b_j < 2^B with A + B <= 60.7 (one operand may be un-normalised).
//
// Values here are NOT canonical, so nothing in this file decides equality: madd29 only applies cheap NECESSARY
// conditions for the exceptional cases of the group law (p = 1 mod 2^29, so k p has low limb k) and reports them; the
// caller hands such a task to the exact 32-bit path (curve.cuh), which is also what fixes the bit-exact output.
#pragma once
#include "curve.cuh"

namespace kh {

#include "field29_asm.inc"

template <class F> struct C29;
template <> struct C29<FpParams> { typedef Fp29C T; };
template <> struct C29<FqParams> { typedef Fq29C T; };

static constexpr u32 MASK29 = 0x1fffffffu;

template <class F>
struct Fe29 {
    u32 v[9];
};

// eight 32-bit words (value X < 2^256 - 2^(256 - SH)) -> nine 29-bit limbs of X << SH, normalised.  SH = 5 turns the
// canonical wire form x R (R = 2^256) into an (unreduced, < 32 p) R'-form of x for free.
template <class F, int SH>
__device__ __forceinline__ Fe29<F> pack29(const Fe<F>& a) {
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i - SH;                    // first bit of limb i within X
        if (bit < 0) { r.v[i] = (a.v[0] << (-bit)) & MASK29; continue; }
        const int w = bit >> 5, s = bit & 31;
        u32 lo = a.v[w], hi = (w + 1 < 8) ? a.v[w + 1] : 0u;
        u32 x = s ? __builtin_amdgcn_alignbit(hi, lo, s) : lo;
        r.v[i] = (i < 8) ? (x & MASK29) : x;
    }
    return r;
}
// nine normalised limbs (value < 2^256) -> eight 32-bit words
template <class F>
__device__ __forceinline__ Fe<F> unpack29(const Fe29<F>& a) {
    Fe<F> r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, i = bit / 29, s = bit - 29 * i;     // word j starts at bit s of limb i
        u32 x = a.v[i] >> s;
        x |= a.v[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < 9) x |= a.v[i + 2] << (58 - s);
        r.v[j] = x;
    }
    return r;
}

#define KH29_CONSTS typedef typename C29<F>::T K; const u32 p1 = K::P1, p2 = K::P2, p3 = K::P3, p4 = K::P4, c22 = 1u << 22, msk = MASK29, pairk = (1u << 29) + 1u

template <class F>
__device__ __forceinline__ Fe29<F> mul29(const Fe29<F>& a, const Fe29<F>& b) {
    Fe29<F> r;
    KH29_CONSTS;
    asm(KH29_MUL_ASM
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(r.v[8])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(a.v[8]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]), "v"(b.v[8]),
          "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(c22), "s"(msk), "v"(pairk)
        : "vcc", "v2", "v3");
    return r;
}
// (a b + c d) / R' mod p with ONE reduction: 212 MADs instead of 2 x 131 (and no subtraction afterwards when the caller
// passes c = K p - c').  Column bound: both products may have one un-normalised operand (limbs < 1.5 * 2^30).
template <class F>
__device__ __forceinline__ Fe29<F> muladd29(const Fe29<F>& a, const Fe29<F>& b, const Fe29<F>& c, const Fe29<F>& d) {
    Fe29<F> r;
    KH29_CONSTS;
    asm(KH29_MULADD_ASM
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(r.v[8])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(a.v[8]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]), "v"(b.v[8]),
          "v"(c.v[0]), "v"(c.v[1]), "v"(c.v[2]), "v"(c.v[3]), "v"(c.v[4]), "v"(c.v[5]), "v"(c.v[6]), "v"(c.v[7]), "v"(c.v[8]),
          "v"(d.v[0]), "v"(d.v[1]), "v"(d.v[2]), "v"(d.v[3]), "v"(d.v[4]), "v"(d.v[5]), "v"(d.v[6]), "v"(d.v[7]), "v"(d.v[8]),
          "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(c22), "s"(msk), "v"(pairk)
        : "vcc", "v2", "v3");
    return r;
}
template <class F>
__device__ __forceinline__ Fe29<F> sqr29(const Fe29<F>& a) {
    Fe29<F> r;
    KH29_CONSTS;
    u32 d1, d2, d3, d4, d5, d6, d7, d8;                    // 2 a_1 .. 2 a_8 (scratch)
    asm(KH29_SQR_ASM
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(r.v[8]),
          "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7), "=&v"(d8)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(a.v[8]),
          "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(c22), "s"(msk), "v"(pairk)
        : "vcc", "v2", "v3");
    return r;
}

// a - b + K p with normalised result: limb-wise a_i + C_i - b_i (C = K p spread over the limbs so that no limb goes
// negative: valid for b_i <= J MASK29, b_8 <= (K p)_8 - J) and one carry sweep.  SPREAD is one of K::s71 ... (K, J in the name).
#define KH29_SUBN(r, a, b, SPREAD)                                                   \
    {                                                                                \
        u32 carry_ = 0;                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 9; i_++) {                           \
            const u32 t_ = (a).v[i_] + SPREAD(i_) + carry_ - (b).v[i_];              \
            if (i_ < 8) { (r).v[i_] = t_ & MASK29; carry_ = t_ >> 29; }              \
            else (r).v[i_] = t_;                                                     \
        }                                                                            \
    }

// wire form (canonical, Montgomery radix 2^256) <-> lazy R'-form
template <class F>
__device__ __forceinline__ Fe29<F> to29(const Fe<F>& a) {          // result < 1.01 p, normalised
    typedef typename C29<F>::T K;
    Fe29<F> k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.v[i] = K::kin(i);
    return mul29<F>(pack29<F, 0>(a), k);
}
template <class F>
__device__ __forceinline__ Fe<F> from29(const Fe29<F>& a) {        // a < 100 p, limbs as mul29 accepts them; result canonical
    typedef typename C29<F>::T K;
    Fe29<F> k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.v[i] = K::kout(i);
    const Fe29<F> t = mul29<F>(a, k);                              // < 2 p < 2^256, normalised
    const Fe<F> w = unpack29<F>(t);
    return cond_sub_p<F>(w.v);
}

template <class F>
struct Acc29 {                 // XYZZ accumulator in lazy R'-form: x < 6 p, y < 4 p (after a madd29: < 1.4 p), zz, zzz < 2 p, all normalised
    Fe29<F> x, y, zz, zzz;
};

// x (limbs 0..7 normalised) == k p as integers?  Runs only behind the one-limb filters below (p = 1 mod 2^29: k p has low limb k),
// i.e. about once per 2^25 additions: without it every such coincidence abandoned the task, and the exact kernel then spent ~0.25 ms of
// one thread's time on it -- in every other 2^20-point MSM (16.7 M additions).
template <class F>
__device__ __forceinline__ bool is_kp29(const Fe29<F>& x, u32 k) {
    typedef typename C29<F>::T K;
    const u32 pl[9] = {1u, K::P1, K::P2, K::P3, K::P4, 0u, 0u, 0u, 1u << 22};
    u64 carry = 0;
    bool same = true;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u64 t = (u64)k * pl[i] + carry;
        const u32 limb = i < 8 ? (u32)(t & MASK29) : (u32)t;
        carry = t >> 29;
        same = same && (limb == x.v[i]);
    }
    return same;
}

// acc += (px, py) with px = 32 X, py = 32 Y (pack29<F, 5> of the canonical affine coordinates; the sign already applied).
// Returns false -- acc untouched -- when an exceptional case of the group law cannot be excluded (acc possibly the
// identity, the points possibly equal or opposite); see the header.  Bounds: tools/gen_field29_asm.py (Madd29Model).
template <class F>
__device__ __forceinline__ bool madd29(Acc29<F>& a, const Fe29<F>& px, const Fe29<F>& py) {
    typedef typename C29<F>::T K;
    if (__builtin_expect(a.zz.v[0] <= 1u, 0) && is_kp29<F>(a.zz, a.zz.v[0])) return false;        // zz in {0, p}: the identity
    const Fe29<F> U2 = mul29<F>(px, a.zz), S2 = mul29<F>(py, a.zzz);
    Fe29<F> P, R;
    KH29_SUBN(P, U2, a.x, K::s71)
    if (__builtin_expect(P.v[0] <= 8u, 0) && is_kp29<F>(P, P.v[0])) return false;                  // P = k p, k <= 8: same x
    KH29_SUBN(R, S2, a.y, K::s51)
    const Fe29<F> PP = sqr29<F>(P);
    const Fe29<F> PPP = mul29<F>(P, PP), Q = mul29<F>(a.x, PP), RR = sqr29<F>(R);
    Fe29<F> sub, rx, t, yn;
#pragma unroll
    for (int i = 0; i < 9; i++) sub.v[i] = PPP.v[i] + 2u * Q.v[i];
    KH29_SUBN(rx, RR, sub, K::s44)
#pragma unroll
    for (int i = 0; i < 9; i++) t.v[i] = Q.v[i] + K::s61(i) - rx.v[i];          // not normalised: limbs < 2^29 + 2^30
#pragma unroll
    for (int i = 0; i < 9; i++) yn.v[i] = K::s51(i) - a.y.v[i];                   // 5 p - y, not normalised
    a.y = muladd29<F>(R, t, yn, PPP);                                           // R (Q - X3) - Y1 PPP, one reduction, < 1.4 p
    a.zz = mul29<F>(a.zz, PP);
    a.zzz = mul29<F>(a.zzz, PPP);
    a.x = rx;
    return true;
}

// a += b, both lazy XYZZ points with the bounds of Acc29 (the full addition add-2008-s, 12M + 2S as 10 products + 2 squarings + one fused
// product pair; limb model: tools/gen_field29_asm.py Madd29Model.add).  Neither operand may be the identity (the callers skip all-zero
// records; a lazy point that went through madd29 / add29 never is one).  Returns false -- a untouched -- when P = U2 - U1 = 0 (mod p)
// cannot be excluded, i.e. the points may be equal or opposite: the caller hands its work item to the exact 32-bit path.
template <class F>
__device__ __forceinline__ bool add29(Acc29<F>& a, const Acc29<F>& b) {
    typedef typename C29<F>::T K;
    const Fe29<F> U1 = mul29<F>(a.x, b.zz), U2 = mul29<F>(b.x, a.zz);
    Fe29<F> P, R;
    KH29_SUBN(P, U2, U1, K::s51)
    if (__builtin_expect(P.v[0] <= 8u, 0) && is_kp29<F>(P, P.v[0])) return false;                  // P = k p: same x
    const Fe29<F> S1 = mul29<F>(a.y, b.zzz), S2 = mul29<F>(b.y, a.zzz);
    KH29_SUBN(R, S2, S1, K::s51)
    const Fe29<F> PP = sqr29<F>(P);
    const Fe29<F> PPP = mul29<F>(P, PP), Q = mul29<F>(U1, PP), RR = sqr29<F>(R);
    Fe29<F> sub, rx, t, yn;
#pragma unroll
    for (int i = 0; i < 9; i++) sub.v[i] = PPP.v[i] + 2u * Q.v[i];
    KH29_SUBN(rx, RR, sub, K::s44)
#pragma unroll
    for (int i = 0; i < 9; i++) t.v[i] = Q.v[i] + K::s61(i) - rx.v[i];          // not normalised
#pragma unroll
    for (int i = 0; i < 9; i++) yn.v[i] = K::s51(i) - S1.v[i];                    // 5 p - S1, not normalised
    a.y = muladd29<F>(R, t, yn, PPP);                                           // R (Q - X3) - S1 PPP
    a.zz = mul29<F>(mul29<F>(a.zz, b.zz), PP);
    a.zzz = mul29<F>(mul29<F>(a.zzz, b.zzz), PPP);
    a.x = rx;
    return true;
}

// A lazy XYZZ point in memory ("B29 record"): 36 words x | y | zz | zzz of nine limbs each, 144 bytes, 16-byte aligned.  The identity is the
// record whose zz limbs are all zero (what to29 makes of the canonical identity; a lazy sum is never congruent to it, see add29).
static constexpr size_t B29_BYTES = 144;
template <class F>
__device__ __forceinline__ Acc29<F> load_b29(const uint8_t* rec) {
    const uint4* q = (const uint4*)rec;
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    Acc29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.x.v[i] = w[i]; r.y.v[i] = w[9 + i]; r.zz.v[i] = w[18 + i]; r.zzz.v[i] = w[27 + i]; }
    return r;
}
template <class F>
__device__ __forceinline__ void store_b29(uint8_t* rec, const Acc29<F>& a) {
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) { w[i] = a.x.v[i]; w[9 + i] = a.y.v[i]; w[18 + i] = a.zz.v[i]; w[27 + i] = a.zzz.v[i]; }
    uint4* q = (uint4*)rec;
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
template <class F>
__device__ __forceinline__ void store_b29_identity(uint8_t* rec) {
    uint4* q = (uint4*)rec;
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4(0u, 0u, 0u, 0u);
}
template <class F>
__device__ __forceinline__ bool is_identity29(const Acc29<F>& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.zz.v[i];
    return o == 0u;
}
// exact point <-> B29 record (the exact kernels that feed or redo lazy work)
template <class F>
__device__ __forceinline__ Acc29<F> xyzz_to29(const Xyzz<F>& p) {
    Acc29<F> r; r.x = to29<F>(p.x); r.y = to29<F>(p.y); r.zz = to29<F>(p.zz); r.zzz = to29<F>(p.zzz); return r;
}
template <class F>
__device__ __forceinline__ Xyzz<F> xyzz_from29(const Acc29<F>& a) {
    Xyzz<F> r; r.x = from29<F>(a.x); r.y = from29<F>(a.y); r.zz = from29<F>(a.zz); r.zzz = from29<F>(a.zzz); return r;
}

}  // namespace kh
