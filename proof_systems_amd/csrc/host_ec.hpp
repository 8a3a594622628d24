// host_ec.hpp -- host-side Pasta field/curve arithmetic used by the product library
// to FINISH device results: fold the per-window sums (Horner), convert XYZZ -> affine
// (one field inversion), and apply blinders h*w + C (SRS::mask_custom,
// poly-commitment/src/ipa.rs:605-622 -- "negligible, stays on host").
// 4 x u64 Montgomery limbs, R = 2^256: the ark-ff in-memory representation.
// Independent of oracle/ (the product never links the oracle).
#pragma once
#include <stdint.h>
#include <string.h>

namespace khost {

typedef uint64_t u64;
typedef unsigned __int128 u128;

struct fe { u64 l[4]; };

struct FieldP {
    fe p; u64 inv; fe one; fe r2;
};

inline const FieldP& field(int id) {   // 0 = Fp, 1 = Fq
    static const FieldP F[2] = {
        {{{0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0ULL, 0x4000000000000000ULL}}, 0x992d30ecffffffffULL,
         {{0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}},
         {{0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL}}},
        {{{0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0ULL, 0x4000000000000000ULL}}, 0x8c46eb20ffffffffULL,
         {{0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}},
         {{0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL}}}};
    return F[id & 1];
}
// curve 0 = Vesta: coordinates Fq(1), scalars Fp(0); curve 1 = Pallas: coordinates Fp(0), scalars Fq(1)
inline int base_field_id(int curve) { return curve == 0 ? 1 : 0; }
inline int scalar_field_id(int curve) { return curve == 0 ? 0 : 1; }

inline bool is_zero(const fe& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool eq(const fe& a, const fe& b) { return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0; }
inline bool geq(const fe& a, const fe& b) {
    for (int i = 3; i >= 0; i--) { if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; }
    return true;
}
inline u64 sub_n(fe& r, const fe& a, const fe& b) {
    u64 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (u64)d; br = (u64)(d >> 64) & 1; }
    return br;
}
inline void add_n(fe& r, const fe& a, const fe& b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (u64)c; c >>= 64; }
}
struct Fld {
    const FieldP& f;
    explicit Fld(int id) : f(field(id)) {}
    fe add(const fe& a, const fe& b) const { fe t; add_n(t, a, b); if (geq(t, f.p)) sub_n(t, t, f.p); return t; }
    fe sub(const fe& a, const fe& b) const { fe t; if (sub_n(t, a, b)) add_n(t, t, f.p); return t; }
    fe neg(const fe& a) const { if (is_zero(a)) return a; fe t; sub_n(t, f.p, a); return t; }
    fe dbl(const fe& a) const { return add(a, a); }
    // Montgomery product, CIOS over 64-bit limbs, using the shape of the Pasta primes p = [p0, p1, 0, 2^62]: the reduction
    // row needs two multiplications (m p0, m p1) and a shift instead of four.  The transcript's Poseidon permutations
    // (1155 products each, strictly sequential between a proof's challenges) are what this is tuned for.
    fe mul(const fe& a, const fe& b) const {
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        const u64 p0 = f.p.l[0], p1 = f.p.l[1];
        for (int i = 0; i < 4; i++) {
            const u64 bi = b.l[i];
            u128 c = (u128)a.l[0] * bi + t0; t0 = (u64)c; c >>= 64;
            c += (u128)a.l[1] * bi + t1; t1 = (u64)c; c >>= 64;
            c += (u128)a.l[2] * bi + t2; t2 = (u64)c; c >>= 64;
            c += (u128)a.l[3] * bi + t3; t3 = (u64)c; c >>= 64;
            c += t4; t4 = (u64)c; const u64 t5 = (u64)(c >> 64);
            const u64 m = t0 * f.inv;
            c = (u128)m * p0 + t0; c >>= 64;
            c += (u128)m * p1 + t1; t0 = (u64)c; c >>= 64;
            c += t2; t1 = (u64)c; c >>= 64;                              // p2 = 0
            c += (u128)(m << 62) + t3; t2 = (u64)c; c >>= 64;            // m * 2^62 = (m >> 2) 2^64 + (m << 62)
            c += (u128)(m >> 2) + t4; t3 = (u64)c; t4 = t5 + (u64)(c >> 64);
        }
        fe r = {{t0, t1, t2, t3}};
        if (t4 || geq(r, f.p)) sub_n(r, r, f.p);
        return r;
    }
    fe sqr(const fe& a) const { return mul(a, a); }
    fe from_mont(const fe& a) const { fe one = {{1, 0, 0, 0}}; return mul(a, one); }
    fe to_mont(const fe& a) const { return mul(a, f.r2); }
    fe inv(const fe& a) const {           // a^(p-2)
        fe e = f.p; e.l[0] -= 2;
        fe acc = f.one, base = a;
        for (int i = 0; i < 255; i++) {
            if ((e.l[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
};

struct xyzz { fe x, y, zz, zzz; };
struct aff { fe x, y; };

struct Crv {
    Fld F;
    explicit Crv(int curve) : F(base_field_id(curve)) {}
    xyzz identity() const { xyzz r; memset(&r, 0, sizeof(r)); return r; }
    bool is_identity(const xyzz& p) const { return is_zero(p.zz); }
    xyzz from_affine(const aff& p) const { return xyzz{p.x, p.y, F.f.one, F.f.one}; }
    xyzz dbl(const xyzz& p) const {
        if (is_identity(p)) return p;
        fe U = F.dbl(p.y), V = F.sqr(U), W = F.mul(U, V), S = F.mul(p.x, V);
        fe X2 = F.sqr(p.x), M = F.add(F.dbl(X2), X2);
        xyzz r;
        r.x = F.sub(F.sub(F.sqr(M), S), S);
        r.y = F.sub(F.mul(M, F.sub(S, r.x)), F.mul(W, p.y));
        r.zz = F.mul(V, p.zz); r.zzz = F.mul(W, p.zzz);
        return r;
    }
    xyzz add(const xyzz& a, const xyzz& b) const {
        if (is_identity(a)) return b;
        if (is_identity(b)) return a;
        fe U1 = F.mul(a.x, b.zz), U2 = F.mul(b.x, a.zz), S1 = F.mul(a.y, b.zzz), S2 = F.mul(b.y, a.zzz);
        fe P = F.sub(U2, U1), R = F.sub(S2, S1);
        if (is_zero(P)) { if (is_zero(R)) return dbl(a); return identity(); }
        fe PP = F.sqr(P), PPP = F.mul(P, PP), Q = F.mul(U1, PP);
        xyzz r;
        r.x = F.sub(F.sub(F.sub(F.sqr(R), PPP), Q), Q);
        r.y = F.sub(F.mul(R, F.sub(Q, r.x)), F.mul(S1, PPP));
        r.zz = F.mul(F.mul(a.zz, b.zz), PP);
        r.zzz = F.mul(F.mul(a.zzz, b.zzz), PPP);
        return r;
    }
    // returns true if the point is the identity (then out is zeroed)
    bool to_affine(const xyzz& p, aff& out) const {
        if (is_identity(p)) { memset(&out, 0, sizeof(out)); return true; }
        // x = X/ZZ, y = Y/ZZZ with a single inversion (of ZZZ); ZZ^3 = ZZZ^2
        fe izzz = F.inv(p.zzz);
        fe izz = F.sqr(F.mul(izzz, p.zz));   // (ZZ/ZZZ)^2 = ZZ^2/ZZZ^2 = ZZ^2/ZZ^3 = 1/ZZ
        out.x = F.mul(p.x, izz);
        out.y = F.mul(p.y, izzz);
        return false;
    }
    // k * P, k a canonical (non-Montgomery) 256-bit integer
    xyzz mul_plain(const xyzz& p, const fe& k) const {
        xyzz acc = identity();
        for (int i = 255; i >= 0; i--) {
            acc = dbl(acc);
            if ((k.l[i >> 6] >> (i & 63)) & 1) acc = add(acc, p);
        }
        return acc;
    }
};

}  // namespace khost
