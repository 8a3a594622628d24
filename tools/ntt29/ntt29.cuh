// ntt29.cuh -- the butterflies of the NTT in the lazy nine-limb arithmetic of field29.cuh.
//
// The transforms are bound by VALU issue on the Montgomery product: 254 instructions on eight 32-bit limbs against
// 166 on nine 29-bit limbs, and the additions / subtractions of a butterfly lose their carry chains (v_add_u32 issues at twice the rate of
// v_addc_co_u32).  What the lazy form costs: values are only known up to a few multiples of p, so every step keeps an explicit magnitude
// invariant, and elements change representation at the pass boundaries (HBM and the twiddle tables stay in wire form).
//
//   wire form      X = x 2^256 mod p, canonical, eight 32-bit words (HBM, twiddle tables)
//   lazy form      any integer congruent to x 2^261 = 32 X, nine limbs; "normalised" = limbs 0..7 < 2^29
//   load           pack29<5>(X) = 32 X < 32 p, then reduce29 -> (0, 2 p)
//   store          mul29(v, pack29<0>(W)) for a wire-form factor W < p:  v W / 2^261 = x w 2^256 -- the product WITH A WIRE-FORM TWIDDLE IS
//                  the conversion back (the inter-pass twiddle, 1/N, or the wire-form one), result < v / 128 + p, then cond_sub_p
//   invariant      every element entering a butterfly step is normalised and < 2.1 p; sums of four (< 8.4 p) are brought back by reduce29
//                  (value - (hi - 1) p with hi = value >> 254: in (0, 2^254 + p)), differences a - b + K p are formed limb-wise with K p
//                  spread over the limbs so that no limb goes negative (Spread29), and every product returns < 1.1 p.
// tests/test_ntt29_model.py replays these helpers limb for limb on Python integers (bounds asserted at the extremes).
#pragma once
#include "field29.cuh"

namespace kh {

template <class F>
struct P29 {                                                // limbs of p = [1, P1, P2, P3, P4, 0, 0, 0, 2^22]
    typedef typename C29<F>::T K;
    __host__ __device__ static constexpr u32 limb(int i) { return i == 0 ? 1u : i == 1 ? K::P1 : i == 2 ? K::P2 : i == 3 ? K::P3 : i == 4 ? K::P4 : i == 8 ? (1u << 22) : 0u; }
};
// KP p spread over the limbs: S_i = (KP p)_i + J 2^29 - [i > 0] J for i < 8, S_8 = (KP p)_8 - J.  Then a_i + S_i - b_i >= 0 for every limb whenever
// b_i <= J (2^29 - 1) (i < 8) and b_8 <= (KP p)_8 - J, and sum_i S_i 2^(29 i) = KP p.
template <class F, int KP, int J>
struct Spread29 {
    u32 v[9];
    constexpr Spread29() : v{} {
        u64 carry = 0;
        for (int k = 0; k < 9; k++) {
            const u64 t = (u64)KP * P29<F>::limb(k) + carry;
            v[k] = k < 8 ? (u32)(t & MASK29) : (u32)t;
            carry = k < 8 ? t >> 29 : 0;
        }
        for (int k = 0; k < 8; k++) v[k] += (u32)J * (1u << 29) - (k > 0 ? (u32)J : 0u);
        v[8] -= (u32)J;
    }
};

template <class F>
__device__ __forceinline__ Fe29<F> add29(const Fe29<F>& a, const Fe29<F>& b) {           // limb-wise, no carries
    Fe29<F> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a - b + KP p, normalised.  a_i < 2^30, b as Spread29 demands.
template <class F, int KP, int J>
__device__ __forceinline__ Fe29<F> sub29(const Fe29<F>& a, const Fe29<F>& b) {
    constexpr Spread29<F, KP, J> S;
    Fe29<F> r;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 t = a.v[i] + S.v[i] + carry - b.v[i];
        if (i < 8) { r.v[i] = t & MASK29; carry = t >> 29; }
        else r.v[i] = t;
    }
    return r;
}
// limbs < 2^32 - 8 and a value < 64 p  ->  normalised, value - (hi - 1) p in (0, 2^254 + p), hi = value >> 254
template <class F>
__device__ __forceinline__ Fe29<F> reduce29(const Fe29<F>& x) {
    Fe29<F> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const u32 t = x.v[i] + c; r.v[i] = t & MASK29; c = t >> 29; }
    const u32 top = x.v[8] + c;
    const int32_t u = (int32_t)(top >> 22) - 1;                          // multiples of p taken off: hi - 1 >= -1
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int64_t t = (int64_t)r.v[i] - (int64_t)u * (int64_t)P29<F>::limb(i) + carry;
        r.v[i] = (u32)t & MASK29; carry = t >> 29;
    }
    int32_t cs = (int32_t)carry;
#pragma unroll
    for (int i = 5; i < 8; i++) { const int32_t t = (int32_t)r.v[i] + cs; r.v[i] = (u32)t & MASK29; cs = t >> 29; }
    r.v[8] = (u32)((int32_t)(top & ((1u << 22) - 1u)) + (int32_t)(1u << 22) + cs);
    return r;
}
// wire form -> lazy form, normalised, < 2 p
template <class F>
__device__ __forceinline__ Fe29<F> lazy_from_wire(const Fe<F>& a) { return reduce29<F>(pack29<F, 5>(a)); }
// lazy value v (limbs < 2^31, v < 100 p) times a WIRE-form factor w: the wire form of the product, canonical
template <class F>
__device__ __forceinline__ Fe<F> wire_times(const Fe29<F>& v, const Fe<F>& w) {
    const Fe29<F> t = mul29<F>(v, pack29<F, 0>(w));                     // < v / 128 + p < 2 p, normalised
    const Fe<F> o = unpack29<F>(t);
    return cond_sub_p<F>(o.v);
}

}  // namespace kh
