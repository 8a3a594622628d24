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
    fe inv_fermat(const fe& a) const {    // a^(p-2): the definition; kept as the cross-check of inv() (tests/cpp/test_host_inv.cpp)
        fe e = f.p; e.l[0] -= 2;
        fe acc = f.one, base = a;
        for (int i = 0; i < 255; i++) {
            if ((e.l[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    // Inversion by batched division steps (Bernstein-Yang "safegcd", variable time -- nothing here is secret-dependent in a way that
    // matters: the prover's transcript values are public): 62 divsteps at a time on the low words of (f, g) give a 2 x 2 transition
    // matrix, which is applied to the full-width (f, g) and, modulo p, to (d, e); ~10 batches of ~62 cheap word steps + 10 small
    // matrix-vector updates against ~315 Montgomery products for a^(p-2): ~1.5 us instead of ~8 us.  An opening round waits for two
    // sequential inversions (L, R -> affine, then 1/u), a proof for ~50.  0 -> 0, like the exponentiation.
    // Input and output in Montgomery form: inv(aR) as integers is a^-1 R^-1, and one product with R^3 makes it a^-1 R.
    struct s62 { int64_t v[5]; };
    static s62 to_s62(const fe& a) {
        const u64 M = ~0ull >> 2;
        s62 r;
        r.v[0] = (int64_t)(a.l[0] & M);
        r.v[1] = (int64_t)(((a.l[0] >> 62) | (a.l[1] << 2)) & M);
        r.v[2] = (int64_t)(((a.l[1] >> 60) | (a.l[2] << 4)) & M);
        r.v[3] = (int64_t)(((a.l[2] >> 58) | (a.l[3] << 6)) & M);
        r.v[4] = (int64_t)(a.l[3] >> 56);
        return r;
    }
    fe inv(const fe& a) const {
        typedef __int128 i128;
        const u64 M62 = ~0ull >> 2;
        const s62 P = to_s62(f.p);
        // p^-1 mod 2^62 (Newton: each step doubles the number of correct low bits; p is odd)
        u64 pinv = f.p.l[0];
        for (int k = 0; k < 6; k++) pinv *= 2 - f.p.l[0] * pinv;
        pinv &= M62;
        s62 F = P, G = to_s62(a), D = {{0, 0, 0, 0, 0}}, E = {{1, 0, 0, 0, 0}};
        int64_t eta = -1;                                   // eta = -delta of the paper, delta = 1 at the start
        for (int round = 0; round < 16; round++) {          // <= 12 rounds of 62 steps suffice for 256-bit inputs (741 steps); 16 is slack
            // ---- 62 division steps on the low words
            u64 u = 1, v = 0, q = 0, r = 1, fl = (u64)F.v[0] | ((u64)F.v[1] << 62), gl = (u64)G.v[0] | ((u64)G.v[1] << 62);
            for (int i = 62;;) {
                const int zeros = __builtin_ctzll(gl | (~0ull << i));
                gl >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
                if (i == 0) break;
                if (eta < 0) {                              // delta > 0 and g odd: (f, g) <- (g, -f)
                    eta = -eta;
                    u64 t = fl; fl = gl; gl = 0 - t;
                    t = u; u = q; q = 0 - t;
                    t = v; v = r; r = 0 - t;
                }
                gl += fl; q += u; r += v;                   // g <- g + f (both odd): even, halved by the next shift
            }
            const int64_t U = (int64_t)u, V = (int64_t)v, Q = (int64_t)q, R = (int64_t)r;
            // ---- (d, e) <- T (d, e) / 2^62 mod p, kept in (-2p, p)
            {
                const int64_t sd = D.v[4] >> 63, se = E.v[4] >> 63;
                int64_t md = (U & sd) + (V & se), me = (Q & sd) + (R & se);
                i128 cd = (i128)U * D.v[0] + (i128)V * E.v[0], ce = (i128)Q * D.v[0] + (i128)R * E.v[0];
                md -= (int64_t)((pinv * (u64)cd + (u64)md) & M62);
                me -= (int64_t)((pinv * (u64)ce + (u64)me) & M62);
                cd += (i128)P.v[0] * md; ce += (i128)P.v[0] * me;
                cd >>= 62; ce >>= 62;                       // the low 62 bits are zero by the choice of md, me
                for (int k = 1; k < 5; k++) {
                    cd += (i128)U * D.v[k] + (i128)V * E.v[k] + (i128)P.v[k] * md;
                    ce += (i128)Q * D.v[k] + (i128)R * E.v[k] + (i128)P.v[k] * me;
                    D.v[k - 1] = (int64_t)((u64)cd & M62); cd >>= 62;
                    E.v[k - 1] = (int64_t)((u64)ce & M62); ce >>= 62;
                }
                D.v[4] = (int64_t)cd; E.v[4] = (int64_t)ce;
            }
            // ---- (f, g) <- T (f, g) / 2^62 (exact)
            {
                i128 cf = (i128)U * F.v[0] + (i128)V * G.v[0], cg = (i128)Q * F.v[0] + (i128)R * G.v[0];
                cf >>= 62; cg >>= 62;
                for (int k = 1; k < 5; k++) {
                    cf += (i128)U * F.v[k] + (i128)V * G.v[k];
                    cg += (i128)Q * F.v[k] + (i128)R * G.v[k];
                    F.v[k - 1] = (int64_t)((u64)cf & M62); cf >>= 62;
                    G.v[k - 1] = (int64_t)((u64)cg & M62); cg >>= 62;
                }
                F.v[4] = (int64_t)cf; G.v[4] = (int64_t)cg;
            }
            if ((G.v[0] | G.v[1] | G.v[2] | G.v[3] | G.v[4]) == 0) break;
        }
        // f = +-gcd.  gcd = p only for a = 0 (p is prime): the inverse of 0 is reported as 0, like a^(p-2)
        const bool fneg = F.v[4] < 0;
        {
            s62 T = F;
            if (fneg) { int64_t c = 0; for (int k = 0; k < 5; k++) { int64_t x = -T.v[k] + c; if (k < 4) { c = x >> 62; x &= (int64_t)M62; } T.v[k] = x; } }
            if (!(T.v[0] == 1 && (T.v[1] | T.v[2] | T.v[3] | T.v[4]) == 0)) { fe z = {{0, 0, 0, 0}}; return z; }
        }
        // d (or -d if f = -1) into [0, p): d is in (-2p, p)
        s62 X = D;
        if (fneg) for (int k = 0; k < 5; k++) X.v[k] = -X.v[k];
        for (int pass = 0; pass < 3; pass++) {              // carry-normalise, then add p while negative
            int64_t c = 0;
            for (int k = 0; k < 4; k++) { const int64_t x = X.v[k] + c; c = x >> 62; X.v[k] = x & (int64_t)M62; }
            X.v[4] += c;
            if (X.v[4] >= 0) break;
            for (int k = 0; k < 5; k++) X.v[k] += P.v[k];
        }
        fe t;
        t.l[0] = (u64)X.v[0] | ((u64)X.v[1] << 62);
        t.l[1] = ((u64)X.v[1] >> 2) | ((u64)X.v[2] << 60);
        t.l[2] = ((u64)X.v[2] >> 4) | ((u64)X.v[3] << 58);
        t.l[3] = ((u64)X.v[3] >> 6) | ((u64)X.v[4] << 56);
        while (geq(t, f.p)) sub_n(t, t, f.p);
        return mul(t, mul(f.r2, f.r2));                     // a^-1 R^-1 (as an integer) x R^3 -> Montgomery form of a^-1
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
