// host_srs.cpp -- SRS::create on the host (poly-commitment/src/ipa.rs:751-778): the product's
// own generator of the monomial basis g_i and the blinding base h, so that a caller (and
// bench.py) can obtain SRS::<G>::create(depth).g without the reference binary:
//   g_i = to_group(pack248(Blake2b-512(be32(i))))     h = to_group(pack248(Blake2b-512("srs_misc" || be32(0))))
// with to_group = Shallue-van de Woestijne map for y^2 = x^3 + 5 (groupmap/src/lib.rs:74-189)
// and the Tonelli-Shanks root ark-ff returns (no sign normalisation).
// Parity: tests pin it against the digests of srs/{vesta,pallas}.srs in tests/golden/.
#include <thread>
#include <vector>

#include "common.hpp"
#include "host_ec.hpp"

namespace {
using khost::fe;
using khost::Fld;
typedef uint64_t u64;

// ---- BLAKE2b-512, single block, unkeyed (RFC 7693)
inline u64 rotr(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
void blake2b512(const uint8_t* msg, size_t len, uint8_t out[64]) {
    static const u64 IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                              0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint8_t SG[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    u64 h[8], m[16], v[16];
    uint8_t blk[128] = {0};
    memcpy(blk, msg, len);
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010040ULL;
    for (int i = 0; i < 16; i++) { u64 w = 0; for (int j = 7; j >= 0; j--) w = (w << 8) | blk[8 * i + j]; m[i] = w; }
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= (u64)len; v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, u64 x, u64 y) {
        v[a] += v[b] + x; v[d] = rotr(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 24);
        v[a] += v[b] + y; v[d] = rotr(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; r++) {
        const uint8_t* s = SG[r % 10];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) { u64 w = h[i] ^ v[i] ^ v[i + 8]; for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(w >> (8 * j)); }
}

struct SqrtCtx {
    Fld F; fe root, t_m1_d2, pm1_d2, five;
    explicit SqrtCtx(int fid) : F(fid) {
        fe pm1 = F.f.p; pm1.l[0] -= 1;
        fe T, Tm1;
        for (int i = 0; i < 4; i++) {
            pm1_d2.l[i] = (pm1.l[i] >> 1) | (i < 3 ? pm1.l[i + 1] << 63 : 0);
            T.l[i] = (pm1.l[i] >> 32) | (i < 3 ? pm1.l[i + 1] << 32 : 0);
        }
        Tm1 = T; Tm1.l[0] -= 1;
        for (int i = 0; i < 4; i++) t_m1_d2.l[i] = (Tm1.l[i] >> 1) | (i < 3 ? Tm1.l[i + 1] << 63 : 0);
        fe f5 = {{5, 0, 0, 0}}; five = F.to_mont(f5);
        root = pow(five, T);                 // 2-adic root of unity = generator^T, generator = 5
    }
    fe pow(const fe& a, const fe& e) const {
        fe acc = F.f.one, base = a;
        for (int i = 0; i < 256; i++) { if ((e.l[i >> 6] >> (i & 63)) & 1) acc = F.mul(acc, base); base = F.sqr(base); }
        return acc;
    }
    bool sqrt(const fe& a, fe& out) const {  // Tonelli-Shanks, ark-ff's choice of root
        if (khost::is_zero(a)) { out = a; return true; }
        if (!khost::eq(pow(a, pm1_d2), F.f.one)) return false;
        fe z = root, w = pow(a, t_m1_d2), x = F.mul(a, w), b = F.mul(x, w);
        int v = 32;
        while (!khost::eq(b, F.f.one)) {
            int k = 0; fe t = b;
            while (!khost::eq(t, F.f.one)) { t = F.sqr(t); k++; }
            w = z;
            for (int i = 0; i < v - k - 1; i++) w = F.sqr(w);
            z = F.sqr(w); b = F.mul(b, z); x = F.mul(x, w); v = k;
        }
        out = x;
        return true;
    }
};

struct GroupMap {          // BWParameters::setup for a = 0, b = 5: u = 1 (groupmap/src/lib.rs:134-165)
    SqrtCtx S; fe u, fu, c1, s, c2;
    explicit GroupMap(int curve) : S(khost::base_field_id(curve)) {
        const Fld& F = S.F;
        u = F.f.one; fu = F.add(F.f.one, S.five);
        fe two = F.add(F.f.one, F.f.one), three = F.add(two, F.f.one);
        c2 = F.inv(three);
        S.sqrt(F.neg(three), s);
        c1 = F.mul(F.sub(s, u), F.inv(two));
    }
    void to_group(const fe& t, fe& x, fe& y) const {
        const Fld& F = S.F;
        fe t2 = F.sqr(t), tpf = F.add(t2, fu), ai = F.mul(tpf, t2);
        fe alpha; if (khost::is_zero(ai)) memset(&alpha, 0, sizeof(alpha)); else alpha = F.inv(ai);
        fe xs[3];
        xs[0] = F.sub(c1, F.mul(F.mul(F.sqr(t2), alpha), s));
        xs[1] = F.sub(F.neg(u), xs[0]);
        xs[2] = F.sub(u, F.mul(F.mul(F.sqr(tpf), F.mul(alpha, tpf)), c2));
        for (int k = 0; k < 3; k++) {
            fe rhs = F.add(F.mul(F.sqr(xs[k]), xs[k]), S.five);
            if (S.sqrt(rhs, y)) { x = xs[k]; return; }
        }
        abort();
    }
    void point_of_random_bytes(const uint8_t* rb, u64* out_xy) const {   // ipa.rs:234-265
        fe t; memset(&t, 0, sizeof(t));
        for (int i = 0; i < 31; i++)
            for (int j = 0; j < 8; j++)
                if ((rb[i] >> j) & 1) { int bit = 247 - (8 * i + j); t.l[bit >> 6] |= (u64)1 << (bit & 63); }
        t = S.F.to_mont(t);
        fe x, y; to_group(t, x, y);
        memcpy(out_xy, &x, 32); memcpy(out_xy + 4, &y, 32);
    }
};
const GroupMap& group_map(int curve) { static const GroupMap M[2] = {GroupMap(0), GroupMap(1)}; return M[curve & 1]; }
}  // namespace

namespace kh {
// GENERATOR^((p-1)/3), GENERATOR = 5: a primitive cube root of unity, Montgomery form (poseidon/src/sponge.rs:110-114)
void endo_coefficient(int field, uint64_t out[4]) {
    SqrtCtx S(field);
    fe pm1 = S.F.f.p; pm1.l[0] -= 1;
    fe e; unsigned __int128 rem = 0;
    for (int i = 3; i >= 0; i--) { unsigned __int128 cur = (rem << 64) | pm1.l[i]; e.l[i] = (u64)(cur / 3); rem = cur % 3; }
    fe r = S.pow(S.five, e);
    memcpy(out, &r, 32);
}
// (endo_q, endo_r) of ipa.rs:214-231: phi(P) = (endo_q x, y) = [endo_r] P. endo_q is the base field's coefficient as is;
// endo_r is the scalar field's coefficient or its square, whichever acts as phi. The reference decides on the curve
// generator; the eigenvalue of phi is the same on every point of the prime-order group, so the first curve point with
// x = 1, 2, 3, ... decides here (no generator constant needed on this side).
void curve_endos(int curve, uint64_t endo_q[4], uint64_t endo_r[4]) {
    const int bf = khost::base_field_id(curve), sf = khost::scalar_field_id(curve);
    SqrtCtx SB(bf); Fld FS(sf);
    const Fld& FB = SB.F;
    fe eq, er;
    endo_coefficient(bf, eq.l); endo_coefficient(sf, er.l);
    khost::aff g; g.x = FB.f.one;
    for (;;) {
        fe rhs = FB.add(FB.mul(FB.sqr(g.x), g.x), SB.five);
        if (SB.sqrt(rhs, g.y)) break;
        g.x = FB.add(g.x, FB.f.one);
    }
    khost::Crv crv(curve);
    khost::aff lhs;
    crv.to_affine(crv.mul_plain(crv.from_affine(g), FS.from_mont(er)), lhs);
    if (!(khost::eq(lhs.x, FB.mul(g.x, eq)) && khost::eq(lhs.y, g.y))) er = FS.sqr(er);
    memcpy(endo_q, &eq, 32); memcpy(endo_r, &er, 32);
}
// ScalarChallenge::to_field_with_length(128, endo) (poseidon/src/sponge.rs:190-226): a = b = 2; for the 64 two-bit
// chunks high to low: a, b doubled, s = +-1 by bit 2i, added to b if bit 2i+1 is clear, else to a; a * endo + b
void scalar_challenge_to_field(int field, const uint64_t chal[2], const uint64_t endo[4], uint64_t out[4]) {
    Fld F(field);
    fe a = F.add(F.f.one, F.f.one), b = a, e;
    memcpy(&e, endo, 32);
    const fe one = F.f.one, mone = F.neg(F.f.one);
    for (int i = 63; i >= 0; i--) {
        a = F.dbl(a); b = F.dbl(b);
        const u64 w = chal[i >> 5]; const int sh = 2 * (i & 31);
        const fe& s = ((w >> sh) & 1) ? one : mone;
        if ((w >> (sh + 1)) & 1) a = F.add(a, s); else b = F.add(b, s);
    }
    fe r = F.add(F.mul(a, e), b);
    memcpy(out, &r, 32);
}
void host_field_inverse(int field, const uint64_t a[4], uint64_t out[4]) {
    Fld F(field); fe x; memcpy(&x, a, 32); x = F.inv(x); memcpy(out, &x, 32);
}
// out[w] = 2^(c w) P for w < W, affine (one shared inversion): the entries a point occupies in the window tables
void host_window_multiples(int curve, const uint64_t xy[8], int W, int c, uint64_t* out_xy) {
    khost::Crv crv(curve); const Fld& F = crv.F;
    khost::aff p; memcpy(&p, xy, 64);
    std::vector<khost::xyzz> v(W);
    v[0] = crv.from_affine(p);
    for (int w = 1; w < W; w++) { v[w] = v[w - 1]; for (int k = 0; k < c; k++) v[w] = crv.dbl(v[w]); }
    // prefix products of zzz, one inversion, then 1/zzz_w and 1/zz_w = (zz_w / zzz_w)^2
    std::vector<fe> pre(W + 1); pre[0] = F.f.one;
    for (int w = 0; w < W; w++) pre[w + 1] = F.mul(pre[w], v[w].zzz);
    fe inv = F.inv(pre[W]);
    for (int w = W - 1; w >= 0; w--) {
        fe izzz = F.mul(inv, pre[w]); inv = F.mul(inv, v[w].zzz);
        fe izz = F.sqr(F.mul(izzz, v[w].zz));
        fe x = F.mul(v[w].x, izz), y = F.mul(v[w].y, izzz);
        memcpy(out_xy + 8 * w, &x, 32); memcpy(out_xy + 8 * w + 4, &y, 32);
    }
}
}  // namespace kh

extern "C" {
int kh_srs_generate(int curve, size_t start, size_t count, uint64_t* out_xy, int threads) {
    KH_REQUIRE(curve == KH_CURVE_VESTA || curve == KH_CURVE_PALLAS, "unknown curve id %d", curve);
    KH_REQUIRE(out_xy || count == 0, "null output");
    KH_REQUIRE(start + count <= ((size_t)1 << 32), "SRS index must fit u32 (ipa.rs:758)");
    const GroupMap& M = group_map(curve);
    if (threads < 1) threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if ((size_t)threads > count) threads = count ? (int)count : 1;
    auto work = [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            uint32_t idx = (uint32_t)(start + i);
            uint8_t msg[4] = {(uint8_t)(idx >> 24), (uint8_t)(idx >> 16), (uint8_t)(idx >> 8), (uint8_t)idx};
            uint8_t dig[64]; blake2b512(msg, 4, dig);
            M.point_of_random_bytes(dig, out_xy + 8 * i);
        }
    };
    if (threads == 1) { work(0, count); return KH_OK; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(work, count * t / threads, count * (t + 1) / threads);
    for (auto& t : th) t.join();
    return KH_OK;
}
// GroupMap::to_group (groupmap/src/lib.rs:167-189) of a base-field element: the U base of SRS::open / verify
// (ipa.rs:909-913: u_base = group_map.to_group(sponge.challenge_fq()))
int kh_group_map_to_group(int curve, const uint64_t t[4], uint64_t out_xy[8]) {
    KH_REQUIRE(curve == KH_CURVE_VESTA || curve == KH_CURVE_PALLAS, "unknown curve id %d", curve);
    KH_REQUIRE(t && out_xy, "null argument");
    fe tt, x, y; memcpy(&tt, t, 32);
    group_map(curve).to_group(tt, x, y);
    memcpy(out_xy, &x, 32); memcpy(out_xy + 4, &y, 32);
    return KH_OK;
}
int kh_srs_h(int curve, uint64_t out_xy[8]) {
    KH_REQUIRE(curve == KH_CURVE_VESTA || curve == KH_CURVE_PALLAS, "unknown curve id %d", curve);
    KH_REQUIRE(out_xy, "null output");
    uint8_t msg[12] = {'s', 'r', 's', '_', 'm', 'i', 's', 'c', 0, 0, 0, 0};
    uint8_t dig[64]; blake2b512(msg, 12, dig);
    group_map(curve).point_of_random_bytes(dig, out_xy);
    return KH_OK;
}
}  // extern "C"
