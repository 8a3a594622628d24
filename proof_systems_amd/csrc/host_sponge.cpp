// host_sponge.cpp -- the Fiat-Shamir sponges of the Kimchi prover / verifier, host side.
//
// The transcript is the one strictly sequential piece of ProverProof::create (kimchi/src/prover.rs:277-1263): every
// challenge depends on the commitments before it.  It stays on the host next to the caller, exactly as in the reference;
// it is in the library so that a device-resident prover loop (proof_systems_amd/prover.py, or the Rust shim) gets its
// challenges without a Python / FFI big-integer detour in the middle of the timed region.  Restates
//   ArithmeticSponge        poseidon/src/poseidon.rs:60-175, permutation.rs:30-116 (PlonkSpongeConstantsKimchi:
//                           width 3, rate 2, 55 full rounds, x^7, no initial round-key addition, constants.rs:29-41)
//   DefaultFqSponge         poseidon/src/sponge.rs:228-412 (absorb_g / absorb_fq / absorb_fr / challenge / digest)
//   DefaultFrSponge         poseidon/src/sponge.rs:240-282, kimchi/src/plonk_sponge.rs:36-57
// Field elements cross the C ABI as 4 x u64 Montgomery limbs like everywhere else; 128-bit challenges as 2 x u64.
// Checked against the oracle's sponge (itself pinned on poseidon/tests/test_vectors/kimchi.json and on the reference's
// opening-proof bytes) in tests/test_sponge.py -- runs without a GPU.
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/kimchi_hip.h"
#include "host_ec.hpp"

namespace kh { void set_error(const char* fmt, ...); }

namespace {
#include "poseidon_params.inc"

// Lazy-reduced Poseidon permutation: values stay in [0, 2p) inside x^7 (redc without the conditional subtraction: inputs below 2p
// give outputs below (2 + tiny) p), the MDS row is ONE reduction of the 512-bit sum of its three products, one conditional subtraction
// of 2p per state element per round.  Same residues as the plain version, canonical state on exit.
template <int FID> struct FastPerm {
    typedef khost::u64 u64; typedef khost::u128 u128; typedef khost::fe fe;
    static constexpr u64 P0 = FID == 0 ? 0x992d30ed00000001ULL : 0x8c46eb2100000001ULL;
    static constexpr u64 P1 = FID == 0 ? 0x224698fc094cf91bULL : 0x224698fc0994a8ddULL;
    static constexpr u64 INV = FID == 0 ? 0x992d30ecffffffffULL : 0x8c46eb20ffffffffULL;
    static constexpr u64 D0 = P0 << 1, D1 = (P1 << 1) | (P0 >> 63), D2 = P1 >> 63, D3 = 0x8000000000000000ULL;   // 2p
    static inline __attribute__((always_inline)) void mul_wide(const u64 a[4], const u64 b[4], u64 t[8]) {
        u128 c = (u128)a[0] * b[0]; t[0] = (u64)c; c >>= 64;
        c += (u128)a[1] * b[0]; t[1] = (u64)c; c >>= 64;
        c += (u128)a[2] * b[0]; t[2] = (u64)c; c >>= 64;
        c += (u128)a[3] * b[0]; t[3] = (u64)c; t[4] = (u64)(c >> 64);
#pragma GCC unroll 3
        for (int i = 1; i < 4; i++) {
            c = (u128)a[0] * b[i] + t[i]; t[i] = (u64)c; c >>= 64;
            c += (u128)a[1] * b[i] + t[i + 1]; t[i + 1] = (u64)c; c >>= 64;
            c += (u128)a[2] * b[i] + t[i + 2]; t[i + 2] = (u64)c; c >>= 64;
            c += (u128)a[3] * b[i] + t[i + 3]; t[i + 3] = (u64)c; t[i + 4] = (u64)(c >> 64);
        }
    }
    static inline __attribute__((always_inline)) void sqr_wide(const u64 a[4], u64 t[8]) {
        // off-diagonal products
        u128 c = (u128)a[0] * a[1]; u64 o1 = (u64)c; c >>= 64;
        c += (u128)a[0] * a[2]; u64 o2 = (u64)c; c >>= 64;
        c += (u128)a[0] * a[3]; u64 o3 = (u64)c; u64 o4 = (u64)(c >> 64);
        c = (u128)a[1] * a[2] + o3; o3 = (u64)c; c >>= 64;
        c += (u128)a[1] * a[3] + o4; o4 = (u64)c; u64 o5 = (u64)(c >> 64);
        c = (u128)a[2] * a[3] + o5; o5 = (u64)c; u64 o6 = (u64)(c >> 64);
        // double
        const u64 o7 = o6 >> 63;
        o6 = (o6 << 1) | (o5 >> 63); o5 = (o5 << 1) | (o4 >> 63); o4 = (o4 << 1) | (o3 >> 63);
        o3 = (o3 << 1) | (o2 >> 63); o2 = (o2 << 1) | (o1 >> 63); o1 <<= 1;
        // diagonals
        c = (u128)a[0] * a[0]; t[0] = (u64)c; c >>= 64;
        c += o1; t[1] = (u64)c; c >>= 64;
        c += (u128)a[1] * a[1] + o2; t[2] = (u64)c; c >>= 64;
        c += o3; t[3] = (u64)c; c >>= 64;
        c += (u128)a[2] * a[2] + o4; t[4] = (u64)c; c >>= 64;
        c += o5; t[5] = (u64)c; c >>= 64;
        c += (u128)a[3] * a[3] + o6; t[6] = (u64)c; c >>= 64;
        c += o7; t[7] = (u64)c;
    }
    // t / 2^256 mod p, in [0, t / 2^256 + p)
    static inline __attribute__((always_inline)) void redc(u64 t[8], u64 r[4]) {
        u64 pc = 0;
#pragma GCC unroll 4
        for (int i = 0; i < 4; i++) {
            const u64 m = t[i] * INV;
            u128 c = (u128)m * P0 + t[i]; c >>= 64;
            c += (u128)m * P1 + t[i + 1]; t[i + 1] = (u64)c; c >>= 64;
            c += t[i + 2]; t[i + 2] = (u64)c; c >>= 64;
            c += (u128)(m << 62) + t[i + 3]; t[i + 3] = (u64)c; c >>= 64;
            c += (u128)(m >> 2) + t[i + 4] + pc; t[i + 4] = (u64)c; pc = (u64)(c >> 64);
        }
        r[0] = t[4]; r[1] = t[5]; r[2] = t[6]; r[3] = t[7];
    }
    static inline __attribute__((always_inline)) void pow7(const u64 x[4], u64 out[4]) {
        u64 t[8], x2[4], x4[4], x6[4];
        sqr_wide(x, t); redc(t, x2);
        sqr_wide(x2, t); redc(t, x4);
        mul_wide(x4, x2, t); redc(t, x6);
        mul_wide(x6, x, t); redc(t, out);
    }
    static inline __attribute__((always_inline)) void add8(u64 t[8], const u64 o[8]) {
        u128 c = 0;
#pragma GCC unroll 8
        for (int i = 0; i < 8; i++) { c += (u128)t[i] + o[i]; t[i] = (u64)c; c >>= 64; }
    }
    // Two builds of the same body: with BMI2 / ADX (mulx keeps the flags out of the multiply chains: 20 against 26 us per permutation
    // on the build host, same source) when the CPU has them -- checked at run time, so the library still loads anywhere.
    __attribute__((target("bmi2,adx"))) static void permute_mulx(fe s[3], const fe mds[3][3], const fe rc[55][3]) { permute_body(s, mds, rc); }
    static void permute(fe s[3], const fe mds[3][3], const fe rc[55][3]) {
        static const bool mulx = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
        if (mulx) permute_mulx(s, mds, rc); else permute_body(s, mds, rc);
    }
    static inline __attribute__((always_inline)) void permute_body(fe s[3], const fe mds[3][3], const fe rc[55][3]) {
        u64 x[3][4];
        for (int i = 0; i < 3; i++) for (int k = 0; k < 4; k++) x[i][k] = s[i].l[k];
        for (int r = 0; r < 55; r++) {
            u64 y[3][4];
            pow7(x[0], y[0]); pow7(x[1], y[1]); pow7(x[2], y[2]);
            for (int i = 0; i < 3; i++) {
                u64 t[8], o[8], v[4];
                mul_wide(mds[i][0].l, y[0], t);
                mul_wide(mds[i][1].l, y[1], o); add8(t, o);
                mul_wide(mds[i][2].l, y[2], o); add8(t, o);
                redc(t, v);                                            // < 2.6 p
                u128 c = (u128)v[0] + rc[r][i].l[0]; v[0] = (u64)c; c >>= 64;    // < 3.6 p < 2^256
                c += (u128)v[1] + rc[r][i].l[1]; v[1] = (u64)c; c >>= 64;
                c += (u128)v[2] + rc[r][i].l[2]; v[2] = (u64)c; c >>= 64;
                c += (u128)v[3] + rc[r][i].l[3]; v[3] = (u64)c;
                // conditional subtraction of 2p: below 2p afterwards
                u128 d = (u128)v[0] - D0; const u64 e0 = (u64)d; u64 br = (u64)(d >> 64) & 1;
                d = (u128)v[1] - D1 - br; const u64 e1 = (u64)d; br = (u64)(d >> 64) & 1;
                d = (u128)v[2] - D2 - br; const u64 e2 = (u64)d; br = (u64)(d >> 64) & 1;
                d = (u128)v[3] - D3 - br; const u64 e3 = (u64)d; br = (u64)(d >> 64) & 1;
                x[i][0] = br ? v[0] : e0; x[i][1] = br ? v[1] : e1; x[i][2] = br ? v[2] : e2; x[i][3] = br ? v[3] : e3;
            }
        }
        for (int i = 0; i < 3; i++) {                                   // canonical on exit
            u128 d = (u128)x[i][0] - P0; const u64 e0 = (u64)d; u64 br = (u64)(d >> 64) & 1;
            d = (u128)x[i][1] - P1 - br; const u64 e1 = (u64)d; br = (u64)(d >> 64) & 1;
            d = (u128)x[i][2] - br; const u64 e2 = (u64)d; br = (u64)(d >> 64) & 1;
            d = (u128)x[i][3] - 0x4000000000000000ULL - br; const u64 e3 = (u64)d; br = (u64)(d >> 64) & 1;
            s[i].l[0] = br ? x[i][0] : e0; s[i].l[1] = br ? x[i][1] : e1; s[i].l[2] = br ? x[i][2] : e2; s[i].l[3] = br ? x[i][3] : e3;
        }
    }
};


// The same permutation with AVX-512 IFMA (vpmadd52luq / vpmadd52huq): the three state elements in three 64-bit lanes, five 52-bit limbs each, Montgomery
// products modulo 2^260.  x^7 of all three elements is FOUR vector products instead of twelve scalar ones, an MDS row + round constant for all three rows
// is three vector products accumulated in one 520-bit sum and ONE reduction (the constant rides in the sum's high half).  R' = 2^260 leaves five spare
// bits above the 255-bit prime: with inputs below B1 p and B2 p a product is below (B1 B2 / 32 + 1) p, so nothing is ever subtracted inside the loop --
// a state element stays below 2.7 p, x^2 < 1.3 p, x^4, x^6, x^7 < 1.2 p.  Entry / exit: one product each with 2^264 resp. 2^256 (both mod p) moves the
// caller's R = 2^256 Montgomery form to R' and back; canonical limbs on exit, the same residues as FastPerm (tests/test_sponge.py runs both).
// On the GPU box's EPYC 9575F a permutation costs ~5 instead of ~9.5 us; a proof makes ~120 of them, 32 inside the opening's 16 round trips.
#if defined(__x86_64__)
#include <immintrin.h>
#define KH_IFMA __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq"), always_inline)) static inline
template <int FID> struct IfmaPerm {
    typedef khost::u64 u64; typedef khost::fe fe;
    static constexpr u64 M52 = (1ULL << 52) - 1;
    struct V5 { __m512i l[5]; };
    struct Tables {
        u64 p[5];                       // the prime in 52-bit limbs (p[3] = 0, p[4] = 2^46)
        u64 pinv;                       // -p^-1 mod 2^52
        alignas(64) u64 k_in[5][8], k_out[5][8];          // 2^264 mod p, 2^256 mod p: broadcast
        alignas(64) u64 mds[3][5][8];                     // mds[j]: lane i holds M[i][j] 2^260 mod p
        alignas(64) u64 rc[55][5][8];                     // rc[r]: lane i holds RC[r][i] 2^260 mod p
        bool ready = false;
    };
    static void to52(const u64 a[4], u64 l[5]) {
        l[0] = a[0] & M52; l[1] = ((a[0] >> 52) | (a[1] << 12)) & M52; l[2] = ((a[1] >> 40) | (a[2] << 24)) & M52;
        l[3] = ((a[2] >> 28) | (a[3] << 36)) & M52; l[4] = a[3] >> 16;
    }
    static void from52(const u64 l[5], u64 a[4]) {       // normalized limbs, value < 2^256
        a[0] = l[0] | (l[1] << 52); a[1] = (l[1] >> 12) | (l[2] << 40); a[2] = (l[2] >> 24) | (l[3] << 28); a[3] = (l[3] >> 36) | (l[4] << 16);
    }
    KH_IFMA V5 normalize(__m512i a0, __m512i a1, __m512i a2, __m512i a3, __m512i a4) {
        const __m512i m = _mm512_set1_epi64((long long)M52);
        V5 r;
        a1 = _mm512_add_epi64(a1, _mm512_srli_epi64(a0, 52)); r.l[0] = _mm512_and_si512(a0, m);
        a2 = _mm512_add_epi64(a2, _mm512_srli_epi64(a1, 52)); r.l[1] = _mm512_and_si512(a1, m);
        a3 = _mm512_add_epi64(a3, _mm512_srli_epi64(a2, 52)); r.l[2] = _mm512_and_si512(a2, m);
        a4 = _mm512_add_epi64(a4, _mm512_srli_epi64(a3, 52)); r.l[3] = _mm512_and_si512(a3, m);
        r.l[4] = a4;                                        // the top limb keeps what is above 2^208 (values stay below 2^258)
        return r;
    }
    // one reduction step on the window t0..t5: t += m p with m = -t0 / p mod 2^52, then t0's carry moves up (t0 itself becomes a multiple of 2^52)
#define KH_REDC_STEP(t0, t1, t2, t3, t4, t5)                                                                                  \
    {                                                                                                                         \
        const __m512i m_ = _mm512_madd52lo_epu64(zero, t0, pinv);                                                            \
        t0 = _mm512_madd52lo_epu64(t0, m_, p0); t1 = _mm512_madd52hi_epu64(t1, m_, p0);                                      \
        t1 = _mm512_madd52lo_epu64(t1, m_, p1); t2 = _mm512_madd52hi_epu64(t2, m_, p1);                                      \
        t2 = _mm512_madd52lo_epu64(t2, m_, p2); t3 = _mm512_madd52hi_epu64(t3, m_, p2);                                      \
        t4 = _mm512_add_epi64(t4, _mm512_and_si512(_mm512_slli_epi64(m_, 46), mask));                                        \
        t5 = _mm512_add_epi64(t5, _mm512_srli_epi64(m_, 6));                                                                 \
        t1 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52));                                                                \
    }
    // a b 2^-260 mod p (+ at most p): a, b with normalized limbs 0..3, limb 4 below 2^52
    KH_IFMA V5 mul(const V5& a, const V5& b, const Tables& T) {
        const __m512i zero = _mm512_setzero_si512(), mask = _mm512_set1_epi64((long long)M52), pinv = _mm512_set1_epi64((long long)T.pinv);
        const __m512i p0 = _mm512_set1_epi64((long long)T.p[0]), p1 = _mm512_set1_epi64((long long)T.p[1]), p2 = _mm512_set1_epi64((long long)T.p[2]);
        __m512i t[10];
        for (int k = 0; k < 10; k++) t[k] = zero;
#pragma GCC unroll 5
        for (int i = 0; i < 5; i++) {
#pragma GCC unroll 5
            for (int j = 0; j < 5; j++) {
                t[i + j] = _mm512_madd52lo_epu64(t[i + j], a.l[j], b.l[i]);
                t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], a.l[j], b.l[i]);
            }
        }
        KH_REDC_STEP(t[0], t[1], t[2], t[3], t[4], t[5]);
        KH_REDC_STEP(t[1], t[2], t[3], t[4], t[5], t[6]);
        KH_REDC_STEP(t[2], t[3], t[4], t[5], t[6], t[7]);
        KH_REDC_STEP(t[3], t[4], t[5], t[6], t[7], t[8]);
        KH_REDC_STEP(t[4], t[5], t[6], t[7], t[8], t[9]);
        return normalize(t[5], t[6], t[7], t[8], t[9]);
    }
    KH_IFMA V5 load(const u64 src[5][8]) {
        V5 r;
        for (int k = 0; k < 5; k++) r.l[k] = _mm512_load_si512((const void*)src[k]);
        return r;
    }
    KH_IFMA void permute_body(fe s[3], const Tables& T) {
        alignas(64) u64 buf[5][8];
        memset(buf, 0, sizeof(buf));
        for (int i = 0; i < 3; i++) { u64 l[5]; to52(s[i].l, l); for (int k = 0; k < 5; k++) buf[k][i] = l[k]; }
        V5 x = mul(load(buf), load(T.k_in), T);
        const __m512i zero = _mm512_setzero_si512(), mask = _mm512_set1_epi64((long long)M52), pinv = _mm512_set1_epi64((long long)T.pinv);
        const __m512i p0 = _mm512_set1_epi64((long long)T.p[0]), p1 = _mm512_set1_epi64((long long)T.p[1]), p2 = _mm512_set1_epi64((long long)T.p[2]);
        const __m512i lane[3] = {_mm512_set1_epi64(0), _mm512_set1_epi64(1), _mm512_set1_epi64(2)};
        const V5 mcol[3] = {load(T.mds[0]), load(T.mds[1]), load(T.mds[2])};
        for (int r = 0; r < 55; r++) {
            const V5 x2 = mul(x, x, T), x4 = mul(x2, x2, T), x6 = mul(x4, x2, T), y = mul(x6, x, T);
            // rows of the MDS matrix: sum_j mcol[j] * broadcast(y lane j), the round constant in the high half, one reduction
            __m512i t[10];
            for (int k = 0; k < 5; k++) t[k] = zero;
            for (int k = 0; k < 5; k++) t[5 + k] = _mm512_load_si512((const void*)T.rc[r][k]);
#pragma GCC unroll 3
            for (int j = 0; j < 3; j++) {
                __m512i yb[5];
                for (int k = 0; k < 5; k++) yb[k] = _mm512_permutexvar_epi64(lane[j], y.l[k]);
#pragma GCC unroll 5
                for (int i = 0; i < 5; i++) {
#pragma GCC unroll 5
                    for (int k = 0; k < 5; k++) {
                        t[i + k] = _mm512_madd52lo_epu64(t[i + k], mcol[j].l[k], yb[i]);
                        t[i + k + 1] = _mm512_madd52hi_epu64(t[i + k + 1], mcol[j].l[k], yb[i]);
                    }
                }
            }
            KH_REDC_STEP(t[0], t[1], t[2], t[3], t[4], t[5]);
            KH_REDC_STEP(t[1], t[2], t[3], t[4], t[5], t[6]);
            KH_REDC_STEP(t[2], t[3], t[4], t[5], t[6], t[7]);
            KH_REDC_STEP(t[3], t[4], t[5], t[6], t[7], t[8]);
            KH_REDC_STEP(t[4], t[5], t[6], t[7], t[8], t[9]);
            x = normalize(t[5], t[6], t[7], t[8], t[9]);
        }
        const V5 o = mul(x, load(T.k_out), T);             // back to R = 2^256: below 1.1 p
        for (int k = 0; k < 5; k++) _mm512_store_si512((void*)buf[k], o.l[k]);
        for (int i = 0; i < 3; i++) {
            u64 l[5], a[4];
            for (int k = 0; k < 5; k++) l[k] = buf[k][i];
            from52(l, a);
            fe v = {{a[0], a[1], a[2], a[3]}};
            const khost::FieldP& f = khost::field(FID);
            while (khost::geq(v, f.p)) khost::sub_n(v, v, f.p);
            s[i] = v;
        }
    }
#undef KH_REDC_STEP
    // tables: the constants through the SAME product (x 2^264 2^-260 = x 16), three at a time
    __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq"))) static void convert3(const Tables& T, const fe* e0, const fe* e1, const fe* e2, u64 dst[5][8]) {
        alignas(64) u64 buf[5][8];
        memset(buf, 0, sizeof(buf));
        const fe* es[3] = {e0, e1, e2};
        for (int i = 0; i < 3; i++) { u64 q[5]; to52(es[i]->l, q); for (int k = 0; k < 5; k++) buf[k][i] = q[k]; }
        const V5 o = mul(load(buf), load(T.k_in), T);
        for (int k = 0; k < 5; k++) _mm512_store_si512((void*)dst[k], o.l[k]);
    }
    __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq"))) static void build(Tables& T, const fe mds[3][3], const fe rc[55][3]) {
        const khost::FieldP& f = khost::field(FID);
        khost::Fld F(FID);
        to52(f.p.l, T.p);
        u64 inv = f.p.l[0];                                 // Newton: p^-1 mod 2^64
        for (int k = 0; k < 6; k++) inv *= 2 - f.p.l[0] * inv;
        T.pinv = (0 - inv) & M52;
        fe k_out = f.one, k_in = f.one;                     // 2^256 mod p as an integer; times 2^8
        for (int k = 0; k < 8; k++) k_in = F.dbl(k_in);
        u64 l[5];
        to52(k_in.l, l); for (int k = 0; k < 5; k++) for (int i = 0; i < 8; i++) T.k_in[k][i] = l[k];
        to52(k_out.l, l); for (int k = 0; k < 5; k++) for (int i = 0; i < 8; i++) T.k_out[k][i] = l[k];
        for (int j = 0; j < 3; j++) convert3(T, &mds[0][j], &mds[1][j], &mds[2][j], T.mds[j]);
        for (int r = 0; r < 55; r++) convert3(T, &rc[r][0], &rc[r][1], &rc[r][2], T.rc[r]);
        T.ready = true;
    }
    __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq"))) static void permute(fe s[3], const fe mds[3][3], const fe rc[55][3]) {
        static Tables T;
        static std::once_flag once;
        std::call_once(once, [&] { build(T, mds, rc); });
        permute_body(s, T);
    }
};
#undef KH_IFMA
static bool ifma_usable() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") &&
                           __builtin_cpu_supports("avx512dq") && !(getenv("KH_NO_IFMA") && atoi(getenv("KH_NO_IFMA")) != 0);
    return ok;
}
#else
static bool ifma_usable() { return false; }
#endif


struct Arith {                       // ArithmeticSponge over field `fid`
    int fid;
    khost::fe s[3];
    bool squeezed = false;           // SpongeState::Squeezed(n) / Absorbed(n)
    int n = 0;
    explicit Arith(int f) : fid(f) { memset(s, 0, sizeof(s)); }
    void permute() {
#if defined(__x86_64__)
        if (ifma_usable()) {
            if (fid == 0) IfmaPerm<0>::permute(s, POSEIDON_MDS_FP, POSEIDON_RC_FP);
            else IfmaPerm<1>::permute(s, POSEIDON_MDS_FQ, POSEIDON_RC_FQ);
            return;
        }
#endif
        if (fid == 0) FastPerm<0>::permute(s, POSEIDON_MDS_FP, POSEIDON_RC_FP);
        else FastPerm<1>::permute(s, POSEIDON_MDS_FQ, POSEIDON_RC_FQ);
    }
    void absorb(const khost::fe& x) {
        khost::Fld F(fid);
        if (!squeezed) {
            if (n == 2) { permute(); s[0] = F.add(s[0], x); n = 1; }
            else { s[n] = F.add(s[n], x); n++; }
        } else {
            s[0] = F.add(s[0], x);
            squeezed = false; n = 1;
        }
    }
    khost::fe squeeze() {
        if (squeezed && n < 2) { n++; return s[n - 1]; }
        permute();
        squeezed = true; n = 1;
        return s[0];
    }
};
}  // namespace

struct kh_sponge {
    int kind;                        // 0: Fq-sponge of `curve`, 1: Fr-sponge of `curve`
    int curve;
    Arith sp;
    std::vector<uint64_t> last_squeezed;
    kh_sponge(int k, int c) : kind(k), curve(c), sp(k == 0 ? khost::base_field_id(c) : khost::scalar_field_id(c)) {}
};

extern "C" {

int kh_sponge_new(int kind, int curve, kh_sponge_t** out) {
    if (!out || (kind != KH_SPONGE_FQ && kind != KH_SPONGE_FR) || (curve != KH_CURVE_VESTA && curve != KH_CURVE_PALLAS)) {
        kh::set_error("kh_sponge_new: bad argument"); return KH_E_INVALID;
    }
    *out = new (std::nothrow) kh_sponge(kind, curve);
    if (!*out) { kh::set_error("out of memory"); return KH_E_NOMEM; }
    return KH_OK;
}
int kh_sponge_clone(const kh_sponge_t* s, kh_sponge_t** out) {
    if (!s || !out) { kh::set_error("kh_sponge_clone: null argument"); return KH_E_INVALID; }
    *out = new (std::nothrow) kh_sponge(*s);
    if (!*out) { kh::set_error("out of memory"); return KH_E_NOMEM; }
    return KH_OK;
}
void kh_sponge_free(kh_sponge_t* s) { delete s; }

// FqSponge::absorb_g: (x, y) of every point, (0, 0) for the point at infinity
int kh_sponge_absorb_g(kh_sponge_t* s, const uint64_t* xy, const uint8_t* inf, size_t n) {
    if (!s || s->kind != KH_SPONGE_FQ || (!xy && n)) { kh::set_error("kh_sponge_absorb_g: needs an Fq-sponge and points"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    for (size_t i = 0; i < n; i++) {
        khost::fe x, y;
        if (inf && inf[i]) { memset(&x, 0, 32); memset(&y, 0, 32); }
        else { memcpy(&x, xy + 8 * i, 32); memcpy(&y, xy + 8 * i + 4, 32); }
        s->sp.absorb(x); s->sp.absorb(y);
    }
    return KH_OK;
}
// elements of the sponge's OWN field (FqSponge::absorb_fq; FrSponge::absorb / absorb_multiple)
int kh_sponge_absorb(kh_sponge_t* s, const uint64_t* x, size_t n) {
    if (!s || (!x && n)) { kh::set_error("kh_sponge_absorb: null argument"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    for (size_t i = 0; i < n; i++) { khost::fe v; memcpy(&v, x + 4 * i, 32); s->sp.absorb(v); }
    return KH_OK;
}
// FqSponge::absorb_fr: SCALAR-field elements into the base-field sponge (sponge.rs:337-366): as one base-field element when
// the scalar modulus is the smaller one (Vesta), else as (high 254 bits, low bit) (Pallas)
int kh_sponge_absorb_fr(kh_sponge_t* s, const uint64_t* x, size_t n) {
    if (!s || s->kind != KH_SPONGE_FQ || (!x && n)) { kh::set_error("kh_sponge_absorb_fr: needs an Fq-sponge"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    khost::Fld SF(khost::scalar_field_id(s->curve)), BF(khost::base_field_id(s->curve));
    const bool scalar_smaller = !khost::geq(SF.f.p, BF.f.p);
    for (size_t i = 0; i < n; i++) {
        khost::fe v; memcpy(&v, x + 4 * i, 32);
        khost::fe c = SF.from_mont(v);                        // canonical integer
        if (scalar_smaller) s->sp.absorb(BF.to_mont(c));
        else {
            khost::fe low = {{c.l[0] & 1, 0, 0, 0}}, high;
            for (int k = 0; k < 4; k++) high.l[k] = (c.l[k] >> 1) | (k < 3 ? c.l[k + 1] << 63 : 0);
            s->sp.absorb(BF.to_mont(high)); s->sp.absorb(BF.to_mont(low));
        }
    }
    return KH_OK;
}
static void squeeze_limbs(kh_sponge* s, size_t k, uint64_t* out) {
    while (s->last_squeezed.size() < k) {
        khost::Fld F(s->sp.fid);
        const khost::fe x = F.from_mont(s->sp.squeeze());
        s->last_squeezed.push_back(x.l[0]); s->last_squeezed.push_back(x.l[1]);      // HIGH_ENTROPY_LIMBS = 2
    }
    for (size_t i = 0; i < k; i++) out[i] = s->last_squeezed[i];
    s->last_squeezed.erase(s->last_squeezed.begin(), s->last_squeezed.begin() + k);
}
// the 128-bit challenge of FqSponge::challenge / FrSponge::challenge (CHALLENGE_LENGTH_IN_LIMBS = 2), raw limbs
int kh_sponge_challenge(kh_sponge_t* s, uint64_t chal[2]) {
    if (!s || !chal) { kh::set_error("kh_sponge_challenge: null argument"); return KH_E_INVALID; }
    squeeze_limbs(s, 2, chal);
    return KH_OK;
}
// the same challenge as an element of the curve's SCALAR field (beta, gamma: used as they are, prover.rs:630-633)
int kh_sponge_challenge_field(kh_sponge_t* s, uint64_t out[4]) {
    if (!s || !out) { kh::set_error("kh_sponge_challenge_field: null argument"); return KH_E_INVALID; }
    uint64_t c[2]; squeeze_limbs(s, 2, c);
    khost::Fld SF(khost::scalar_field_id(s->curve));
    const khost::fe v = {{c[0], c[1], 0, 0}};
    const khost::fe m = SF.to_mont(v);
    memcpy(out, &m, 32);
    return KH_OK;
}
// FqSponge::challenge_fq / digest_fq; FrSponge::digest: one element of the sponge's own field
int kh_sponge_squeeze_field(kh_sponge_t* s, uint64_t out[4]) {
    if (!s || !out) { kh::set_error("kh_sponge_squeeze_field: null argument"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    const khost::fe x = s->sp.squeeze();
    memcpy(out, &x, 32);
    return KH_OK;
}
// FqSponge::digest: the squeezed base-field element as a SCALAR-field element, zero when it does not fit (sponge.rs:368-377)
int kh_sponge_digest(kh_sponge_t* s, uint64_t out[4]) {
    if (!s || !out) { kh::set_error("kh_sponge_digest: null argument"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    if (s->kind == KH_SPONGE_FR) { const khost::fe x = s->sp.squeeze(); memcpy(out, &x, 32); return KH_OK; }
    khost::Fld BF(khost::base_field_id(s->curve)), SF(khost::scalar_field_id(s->curve));
    const khost::fe c = BF.from_mont(s->sp.squeeze());
    khost::fe r; memset(&r, 0, 32);
    if (!khost::geq(c, SF.f.p)) r = SF.to_mont(c);
    memcpy(out, &r, 32);
    return KH_OK;
}

}  // extern "C"
