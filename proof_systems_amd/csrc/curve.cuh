// curve.cuh -- Pallas / Vesta group law on the device (y^2 = x^3 + 5, a = 0;
// curves/src/pasta/curves/{pallas,vesta}.rs of the reference), in extended
// Jacobian ("XYZZ") coordinates: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.
// The point at infinity is ZZ == 0.  All exceptional cases (identity operand,
// P + P, P + (-P)) produce the exact group result: parity with the reference is
// bit-exact, so "negligible probability" shortcuts are not allowed.
#pragma once
#include "field.cuh"

namespace kh {

template <class F>
struct Aff {           // affine, never the identity on the device (filtered by the digit pass)
    Fe<F> x, y;
    __device__ __forceinline__ static Aff load(const void* p) {
        Aff r; r.x = Fe<F>::load(p); r.y = Fe<F>::load((const char*)p + 32); return r;
    }
};

template <class F>
struct Xyzz {
    Fe<F> x, y, zz, zzz;
    __device__ __forceinline__ static Xyzz identity() {
        Xyzz r; r.x = Fe<F>::zero(); r.y = Fe<F>::zero(); r.zz = Fe<F>::zero(); r.zzz = Fe<F>::zero(); return r;
    }
    __device__ __forceinline__ static Xyzz from_affine(const Aff<F>& p) {
        Xyzz r; r.x = p.x; r.y = p.y; r.zz = Fe<F>::one(); r.zzz = Fe<F>::one(); return r;
    }
    __device__ __forceinline__ bool is_identity() const { return zz.is_zero(); }
    __device__ __forceinline__ static Xyzz load(const void* p) {
        const char* c = (const char*)p; Xyzz r;
        r.x = Fe<F>::load(c); r.y = Fe<F>::load(c + 32); r.zz = Fe<F>::load(c + 64); r.zzz = Fe<F>::load(c + 96);
        return r;
    }
    __device__ __forceinline__ void store(void* p) const {
        char* c = (char*)p; x.store(c); y.store(c + 32); zz.store(c + 64); zzz.store(c + 96);
    }
};

// 2 * (affine P)  (mdbl-2008-s-1)
template <class F>
__device__ __forceinline__ Xyzz<F> dbl_affine(const Aff<F>& p) {
    Xyzz<F> r;
    Fe<F> U = dbl<F>(p.y);
    Fe<F> V = sqr<F>(U);
    Fe<F> W = mul<F>(U, V);
    Fe<F> S = mul<F>(p.x, V);
    Fe<F> X2 = sqr<F>(p.x);
    Fe<F> M = add<F>(dbl<F>(X2), X2);
    r.x = sub<F>(sub<F>(sqr<F>(M), S), S);
    r.y = sub<F>(mul<F>(M, sub<F>(S, r.x)), mul<F>(W, p.y));
    r.zz = V; r.zzz = W;          // y == 0 -> V == 0 -> identity, as it must be
    return r;
}

// 2 * P  (dbl-2008-s-1)
template <class F>
__device__ __forceinline__ Xyzz<F> dbl(const Xyzz<F>& p) {
    Xyzz<F> r;
    Fe<F> U = dbl<F>(p.y);
    Fe<F> V = sqr<F>(U);
    Fe<F> W = mul<F>(U, V);
    Fe<F> S = mul<F>(p.x, V);
    Fe<F> X2 = sqr<F>(p.x);
    Fe<F> M = add<F>(dbl<F>(X2), X2);
    r.x = sub<F>(sub<F>(sqr<F>(M), S), S);
    r.y = sub<F>(mul<F>(M, sub<F>(S, r.x)), mul<F>(W, p.y));
    r.zz = mul<F>(V, p.zz); r.zzz = mul<F>(W, p.zzz);   // identity in -> identity out (zz = 0)
    return r;
}

// acc + (+-)P with P affine (madd-2008-s, 8M + 2S) -- exact on every input.
template <class F>
__device__ __forceinline__ Xyzz<F> madd(const Xyzz<F>& a, const Aff<F>& p_in, bool negate) {
    Aff<F> p = p_in;
    if (negate) p.y = neg<F>(p.y);
    if (a.is_identity()) return Xyzz<F>::from_affine(p);
    Fe<F> U2 = mul<F>(p.x, a.zz);
    Fe<F> S2 = mul<F>(p.y, a.zzz);
    Fe<F> P = sub<F>(U2, a.x);
    Fe<F> R = sub<F>(S2, a.y);
    if (P.is_zero() && R.is_zero()) return dbl_affine<F>(p);     // same point
    Fe<F> PP = sqr<F>(P);
    Fe<F> PPP = mul<F>(P, PP);
    Fe<F> Q = mul<F>(a.x, PP);
    Xyzz<F> r;
    r.x = sub<F>(sub<F>(sub<F>(sqr<F>(R), PPP), Q), Q);
    r.y = sub<F>(mul<F>(R, sub<F>(Q, r.x)), mul<F>(a.y, PPP));
    r.zz = mul<F>(a.zz, PP);       // P == 0, R != 0 (opposite points) -> zz = 0 = identity
    r.zzz = mul<F>(a.zzz, PPP);
    return r;
}

// a + b, both XYZZ (add-2008-s, 12M + 2S) -- exact on every input.
template <class F>
__device__ __forceinline__ Xyzz<F> add(const Xyzz<F>& a, const Xyzz<F>& b) {
    if (a.is_identity()) return b;
    if (b.is_identity()) return a;
    Fe<F> U1 = mul<F>(a.x, b.zz);
    Fe<F> U2 = mul<F>(b.x, a.zz);
    Fe<F> S1 = mul<F>(a.y, b.zzz);
    Fe<F> S2 = mul<F>(b.y, a.zzz);
    Fe<F> P = sub<F>(U2, U1);
    Fe<F> R = sub<F>(S2, S1);
    if (P.is_zero() && R.is_zero()) return dbl<F>(a);
    Fe<F> PP = sqr<F>(P);
    Fe<F> PPP = mul<F>(P, PP);
    Fe<F> Q = mul<F>(U1, PP);
    Xyzz<F> r;
    r.x = sub<F>(sub<F>(sub<F>(sqr<F>(R), PPP), Q), Q);
    r.y = sub<F>(mul<F>(R, sub<F>(Q, r.x)), mul<F>(S1, PPP));
    r.zz = mul<F>(mul<F>(a.zz, b.zz), PP);
    r.zzz = mul<F>(mul<F>(a.zzz, b.zzz), PPP);
    return r;
}

// k * P, k a canonical (non-Montgomery) 256-bit integer in eight 32-bit words: left-to-right double-and-add
template <class F>
__device__ __forceinline__ Xyzz<F> scalar_mul(const Xyzz<F>& p, const u32 k[8]) {
    Xyzz<F> acc = Xyzz<F>::identity();
    int top = 7;
    while (top > 0 && k[top] == 0) top--;
    for (int w = top; w >= 0; w--) {
        u32 word = k[w];
        for (int b = 31; b >= 0; b--) {
            acc = dbl<F>(acc);
            if ((word >> b) & 1u) acc = add<F>(acc, p);
        }
    }
    return acc;
}

template <class F>
__device__ __forceinline__ Xyzz<F> negate(const Xyzz<F>& a) { Xyzz<F> r = a; r.y = neg<F>(a.y); return r; }

}  // namespace kh
