// coop.cuh -- XYZZ addition by four cooperating lanes ("quad"), for the latency-bound tree phases of the bucket reduction.
//
// In a tree reduction half of the lanes go idle at every level and what is left is one long dependent chain of additions
// per wave; a lone wave issues one VALU instruction per ~5.6 cycles whatever its ILP, so the chain length in
// INSTRUCTIONS is the time.  The 14 products of add-2008-s have depth 5 when spread over four lanes:
//
//   quad layout: lane role r = lane & 3 holds ONE coordinate of each operand (0: X, 1: Y, 2: ZZ, 3: ZZZ), 8 VGPRs
//   instead of 32 per point; a 128-byte XYZZ record is loaded by its quad as four consecutive 32-byte pieces.
//
//   round 1   t1 = a * partner(b)            role0: U1 = X1 ZZ2   role2: U2 = X2 ZZ1   role1: S1 = Y1 ZZZ2   role3: S2 = Y2 ZZZ1
//   round 2   roles 0,1: d = partner(t1) - t1 (P, R), d^2 (PP, RR)        roles 2,3: a * b (ZZ1 ZZ2, ZZZ1 ZZZ2)
//   round 3   role0: PPP = P PP   role1: Q = U1 PP   role2: ZZ3 = ZZ12 PP   role3: PPP (again, locally)
//   round 4   role1: S1 PPP   role3: ZZZ3 = ZZZ12 PPP   role0: X3 = RR - PPP - 2Q (no product)
//   round 5   role1: Y3 = R (Q - X3) - S1 PPP
//
// 5 product rounds + ~120 cross-lane moves against 14 products: a 2.3x shorter chain.  Exactness: identity operands are
// selected through (flags broadcast from the ZZ lane), P = 0 with R != 0 gives ZZ3 = 0 = identity by itself, and the
// doubling case P = R = 0 (equal points) falls back to every lane of the quad running the ordinary dbl() on the
// gathered operand -- rare, wave-uniformly guarded.
#pragma once
#include "curve.cuh"

namespace kh {

// Intra-quad moves are DPP quad permutes (one VALU instruction per word, no LDS round trip -- __shfl compiles to ds_bpermute_b32,
// ~100 cycles of latency in a chain that is nothing but latency).  CTRL = quad_perm control: lane i of every quad reads lane
// ((CTRL >> 2 i) & 3) of the same quad.
static constexpr int QP_BCAST0 = 0x00, QP_BCAST1 = 0x55, QP_BCAST2 = 0xaa, QP_BCAST3 = 0xff, QP_XOR2 = 0x4e;     // [r,r,r,r]; [2,3,0,1]
template <int CTRL>
__device__ __forceinline__ u32 quad_perm(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
// value of `v` in the lane of this quad selected by CTRL (all 64 lanes execute)
template <class F, int CTRL>
__device__ __forceinline__ Fe<F> quad_get(const Fe<F>& v) {
    Fe<F> r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = quad_perm<CTRL>(v.v[k]);
    return r;
}
template <int CTRL>
__device__ __forceinline__ bool quad_flag(bool f) { return quad_perm<CTRL>(f ? 1u : 0u) != 0u; }

// this lane's coordinate of A + B, given its coordinate of A (a) and of B (b)
template <class F>
__device__ __forceinline__ Fe<F> quad_add(const Fe<F>& a, const Fe<F>& b) {
    const int role = (int)(threadIdx.x & 3u);
    const bool a_id = quad_flag<QP_BCAST2>(a.is_zero()), b_id = quad_flag<QP_BCAST2>(b.is_zero());
    // sparse inputs (a witness of small values leaves most buckets empty): when every quad of the wave has an identity operand no product is needed
    if (__ballot(!a_id && !b_id) == 0ull) return b_id ? a : b;
    // round 1
    const Fe<F> t1 = mul<F>(a, quad_get<F, QP_XOR2>(b));
    // round 2
    const Fe<F> d = sub<F>(quad_get<F, QP_XOR2>(t1), t1);              // role0: P, role1: R (roles 2,3: unused)
    const bool lo = role < 2;
    const Fe<F> m2 = mul<F>(lo ? d : a, lo ? d : b);                    // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
    const bool p0 = quad_flag<QP_BCAST0>(d.is_zero()), r0 = quad_flag<QP_BCAST1>(d.is_zero());
    // round 3
    const Fe<F> PPb = quad_get<F, QP_BCAST0>(m2), Pb = quad_get<F, QP_BCAST0>(d), U1b = quad_get<F, QP_BCAST0>(t1);
    Fe<F> x3 = role == 0 ? d : (role == 1 ? U1b : (role == 2 ? m2 : Pb));
    Fe<F> y3 = role == 0 ? m2 : PPb;
    const Fe<F> m3 = mul<F>(x3, y3);                                     // PPP | Q | ZZ3 | PPP
    // round 4
    const Fe<F> PPPb = quad_get<F, QP_BCAST0>(m3), RRb = quad_get<F, QP_BCAST1>(m2), Qb = quad_get<F, QP_BCAST1>(m3);
    const Fe<F> m4 = mul<F>(role == 1 ? t1 : m2, role == 1 ? PPPb : m3);   // role1: S1 PPP, role3: ZZZ3 (roles 0,2: unused)
    const Fe<F> X3 = sub<F>(sub<F>(sub<F>(RRb, PPPb), Qb), Qb);            // meaningful in every lane (all inputs broadcast)
    // round 5
    const Fe<F> m5 = mul<F>(d, sub<F>(m3, X3));                          // role1: R (Q - X3)
    Fe<F> res = role == 0 ? X3 : (role == 1 ? sub<F>(m5, m4) : (role == 2 ? m3 : m4));
    // equal points: the formulas above give 0/0 -- double A instead (rare; every lane of the quad does the whole dbl)
    const bool need_dbl = p0 && r0 && !a_id && !b_id;
    if (__ballot(need_dbl) != 0ull) {
        Xyzz<F> A;
        A.x = quad_get<F, QP_BCAST0>(a); A.y = quad_get<F, QP_BCAST1>(a); A.zz = quad_get<F, QP_BCAST2>(a); A.zzz = quad_get<F, QP_BCAST3>(a);
        if (need_dbl) {
            const Xyzz<F> D = dbl<F>(A);
            res = role == 0 ? D.x : (role == 1 ? D.y : (role == 2 ? D.zz : D.zzz));
        }
    }
    if (b_id) res = a;
    if (a_id) res = b;
    return res;
}
// this lane's coordinate of 2 A (dbl-2008-s-1 on the curve y^2 = x^3 + b), given its coordinate of A: 4 product rounds instead of 9 products
//   round 1   role0: XX = X^2            role1: V = U^2, U = 2 Y
//   round 2   role0: S = X V             role1: W = U V             role2: ZZ3 = V ZZ           role3: MM = M^2, M = 3 XX
//   round 3   role1: W Y                 role3: ZZZ3 = W ZZZ        (role0: X3 = MM - 2 S, no product)
//   round 4   role1: Y3 = M (S - X3) - W Y
// The identity (ZZ = 0) doubles to ZZ3 = ZZZ3 = 0 by itself; the Pasta curves have no point of order two.
template <class F>
__device__ __forceinline__ Fe<F> quad_dbl(const Fe<F>& a) {
    const int role = (int)(threadIdx.x & 3u);
    const Fe<F> U = add<F>(a, a);                                         // meaningful in role 1
    const Fe<F> p1 = role == 1 ? U : a;
    const Fe<F> m1 = mul<F>(p1, p1);                                      // XX | V | - | -
    const Fe<F> Vb = quad_get<F, QP_BCAST1>(m1), XXb = quad_get<F, QP_BCAST0>(m1);
    const Fe<F> M = add<F>(add<F>(XXb, XXb), XXb);                         // every lane
    const Fe<F> x2 = role == 1 ? U : (role == 3 ? M : a);
    const Fe<F> y2 = role == 1 ? m1 : (role == 3 ? M : Vb);
    const Fe<F> m2 = mul<F>(x2, y2);                                      // S | W | ZZ3 | MM
    const Fe<F> Wb = quad_get<F, QP_BCAST1>(m2), Sb = quad_get<F, QP_BCAST0>(m2), MMb = quad_get<F, QP_BCAST3>(m2);
    const Fe<F> X3 = sub<F>(sub<F>(MMb, Sb), Sb);                          // every lane
    const Fe<F> m3 = mul<F>(role == 1 ? m2 : Wb, a);                       // - | W Y | - | ZZZ3
    const Fe<F> m4 = mul<F>(M, sub<F>(Sb, X3));                            // role1: M (S - X3)
    return role == 0 ? X3 : (role == 1 ? sub<F>(m4, m3) : (role == 2 ? m2 : m3));
}
// k A for a canonical 256-bit k (eight 32-bit words), left to right; quads of one wave may hold different k (the DPP moves stay inside the quad)
template <class F>
__device__ __forceinline__ Fe<F> quad_scalar_mul(const Fe<F>& a, const u32 k[8]) {
    Fe<F> acc = Fe<F>::zero();
    int top = 7;
    while (top > 0 && k[top] == 0) top--;
    for (int w = top; w >= 0; w--) {
        const u32 word = k[w];
        for (int b = 31; b >= 0; b--) {
            acc = quad_dbl<F>(acc);
            if ((word >> b) & 1u) acc = quad_add<F>(acc, a);
        }
    }
    return acc;
}
// the quad's piece of a 128-byte XYZZ record (x | y | zz | zzz, 32 bytes each)
template <class F>
__device__ __forceinline__ Fe<F> quad_load(const uint8_t* rec) { return Fe<F>::load(rec + 32 * (threadIdx.x & 3u)); }
template <class F>
__device__ __forceinline__ void quad_store(uint8_t* rec, const Fe<F>& v) { v.store(rec + 32 * (threadIdx.x & 3u)); }
// the same coordinate of the point held `quads` quads further up in the wave; the identity beyond the wave's last quad
template <class BF>
__device__ __forceinline__ Fe<BF> quad_shfl_down(const Fe<BF>& v, int quads) {
    Fe<BF> r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = (u32)__shfl_down((int)v.v[k], 4 * quads, 64);
    if ((int)(threadIdx.x & 63u) + 4 * quads >= 64) r = Fe<BF>::zero();
    return r;
}
template <class F>
__device__ __forceinline__ Fe<F> quad_identity() { return Fe<F>::zero(); }     // all-zero record = identity (ZZ = 0)

}  // namespace kh
