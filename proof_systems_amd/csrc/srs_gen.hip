// srs_gen.hip -- SRS::create on the device (poly-commitment/src/ipa.rs:751-778):
// thread i computes g_i = to_group(pack248(Blake2b-512(be32(i)))) with the Shallue-van de Woestijne map of
// groupmap/src/lib.rs:74-189 and the Tonelli-Shanks root ark-ff returns (no sign normalisation), so the
// result is byte-identical to srs/{vesta,pallas}.srs (pinned by digest in the tests).  ~1300 Montgomery
// products per point: 2^20 points in ~15 ms instead of seconds of host threads.
#include "common.hpp"
#include "field.cuh"
#include "host_ec.hpp"
#include "msm.hpp"

namespace kh {

struct SvdwParams {            // Montgomery limbs (8 x u32 each), base field of the curve
    u32 fu[8], c1[8], s[8], c2[8], five[8], root[8];      // root = 5^T (2-adic root of unity)
    u32 t_m1_d2[8];                                        // (T-1)/2, plain integer, T = (p-1) >> 32
};

__device__ __forceinline__ u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
// BLAKE2b-512 of the 4-byte big-endian index (single block, RFC 7693)
__device__ void blake2b_be32(u32 idx, uint8_t out[64]) {
    const u64 IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    const uint8_t SG[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    u64 h[8], m[16], v[16];
    for (int i = 0; i < 16; i++) m[i] = 0;
    // message bytes: idx big-endian in the first 4 bytes (little-endian word load)
    m[0] = (u64)((idx >> 24) & 0xff) | ((u64)((idx >> 16) & 0xff) << 8) | ((u64)((idx >> 8) & 0xff) << 16) | ((u64)(idx & 0xff) << 24);
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010040ULL;
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= 4ULL; v[14] = ~v[14];
#define KH_B2G(a, b, c, d, x, y)                                                                    \
    v[a] += v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] += v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++) {
        const uint8_t* s = SG[r % 10];
        KH_B2G(0, 4, 8, 12, m[s[0]], m[s[1]]) KH_B2G(1, 5, 9, 13, m[s[2]], m[s[3]])
        KH_B2G(2, 6, 10, 14, m[s[4]], m[s[5]]) KH_B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
        KH_B2G(0, 5, 10, 15, m[s[8]], m[s[9]]) KH_B2G(1, 6, 11, 12, m[s[10]], m[s[11]])
        KH_B2G(2, 7, 8, 13, m[s[12]], m[s[13]]) KH_B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef KH_B2G
    for (int i = 0; i < 8; i++) { u64 w = h[i] ^ v[i] ^ v[i + 8]; for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(w >> (8 * j)); }
}

template <class F>
__device__ __forceinline__ Fe<F> fe_from(const u32 w[8]) { Fe<F> r; for (int i = 0; i < 8; i++) r.v[i] = w[i]; return r; }

// Tonelli-Shanks exactly as ark-ff / the oracle.  Returns false for a non-residue.
template <class F>
__device__ bool fe_sqrt(const Fe<F>& a, const SvdwParams& P, Fe<F>& out) {
    if (a.is_zero()) { out = a; return true; }
    const Fe<F> one = Fe<F>::one();
    // w = a^((T-1)/2)
    Fe<F> w = one;
    int top = 255;
    while (top > 0 && !((P.t_m1_d2[top >> 5] >> (top & 31)) & 1u)) top--;
    for (int i = top; i >= 0; i--) {
        w = sqr<F>(w);
        if ((P.t_m1_d2[i >> 5] >> (i & 31)) & 1u) w = mul<F>(w, a);
    }
    Fe<F> x = mul<F>(a, w);
    Fe<F> b = mul<F>(x, w);              // a^T
    {                                    // residue test: b^(2^31) == 1
        Fe<F> t = b;
        for (int i = 0; i < 31; i++) t = sqr<F>(t);
        if (!(t == one)) return false;
    }
    Fe<F> z = fe_from<F>(P.root);
    int v = 32;
    while (!(b == one)) {
        int k = 0; Fe<F> t = b;
        while (!(t == one)) { t = sqr<F>(t); k++; }
        w = z;
        for (int i = 0; i < v - k - 1; i++) w = sqr<F>(w);
        z = sqr<F>(w);
        b = mul<F>(b, z);
        x = mul<F>(x, w);
        v = k;
    }
    out = x;
    return true;
}

template <class F>
__global__ void __launch_bounds__(128)
k_srs_generate(u32 start, size_t count, SvdwParams P, uint8_t* __restrict__ out_xy) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint8_t dig[64];
    blake2b_be32(start + (u32)i, dig);
    // 31 bytes -> 248 bits, LSB-first inside each byte, read big-endian (ipa.rs:234-259)
    Fe<F> t = Fe<F>::zero();
    for (int by = 0; by < 31; by++)
        for (int j = 0; j < 8; j++)
            if ((dig[by] >> j) & 1) { int bit = 247 - (8 * by + j); t.v[bit >> 5] |= 1u << (bit & 31); }
    t = to_mont<F>(t);
    const Fe<F> one = Fe<F>::one(), fu = fe_from<F>(P.fu), c1 = fe_from<F>(P.c1), s = fe_from<F>(P.s), c2 = fe_from<F>(P.c2), five = fe_from<F>(P.five);
    Fe<F> t2 = sqr<F>(t), tpf = add<F>(t2, fu), ai = mul<F>(tpf, t2);
    Fe<F> alpha = ai.is_zero() ? Fe<F>::zero() : inv<F>(ai);
    Fe<F> xs[3];
    xs[0] = sub<F>(c1, mul<F>(mul<F>(sqr<F>(t2), alpha), s));
    xs[1] = sub<F>(neg<F>(one), xs[0]);                                     // -u - x1, u = 1
    xs[2] = sub<F>(one, mul<F>(mul<F>(sqr<F>(tpf), mul<F>(alpha, tpf)), c2));
    Fe<F> x = Fe<F>::zero(), y = Fe<F>::zero();
    for (int k = 0; k < 3; k++) {
        Fe<F> rhs = add<F>(mul<F>(sqr<F>(xs[k]), xs[k]), five);
        if (fe_sqrt<F>(rhs, P, y)) { x = xs[k]; break; }
    }
    x.store(out_xy + i * 64); y.store(out_xy + i * 64 + 32);
}

int srs_generate_device(Context& C, int curve, size_t start, size_t count, void* out_xy_dev) {
    const int fid = khost::base_field_id(curve);
    khost::Fld F(fid);
    SvdwParams P;
    auto put = [](u32 dst[8], const khost::fe& v) { memcpy(dst, &v, 32); };
    khost::fe five = {{5, 0, 0, 0}}; five = F.to_mont(five);
    khost::fe two = F.add(F.f.one, F.f.one), three = F.add(two, F.f.one);
    khost::fe pm1 = F.f.p; pm1.l[0] -= 1;
    khost::fe T, Tm1, tm1d2;
    for (int i = 0; i < 4; i++) T.l[i] = (pm1.l[i] >> 32) | (i < 3 ? pm1.l[i + 1] << 32 : 0);
    Tm1 = T; Tm1.l[0] -= 1;
    for (int i = 0; i < 4; i++) tm1d2.l[i] = (Tm1.l[i] >> 1) | (i < 3 ? Tm1.l[i + 1] << 63 : 0);
    auto hpow = [&](const khost::fe& a, const khost::fe& e) { khost::fe acc = F.f.one, b = a; for (int i = 0; i < 256; i++) { if ((e.l[i >> 6] >> (i & 63)) & 1) acc = F.mul(acc, b); b = F.sqr(b); } return acc; };
    khost::fe root = hpow(five, T);
    // host Tonelli-Shanks for the one constant sqrt(-3)
    auto hsqrt = [&](const khost::fe& a) {
        khost::fe z = root, w = hpow(a, tm1d2), x = F.mul(a, w), b = F.mul(x, w);
        int v = 32;
        while (!khost::eq(b, F.f.one)) {
            int k = 0; khost::fe t = b;
            while (!khost::eq(t, F.f.one)) { t = F.sqr(t); k++; }
            w = z; for (int i = 0; i < v - k - 1; i++) w = F.sqr(w);
            z = F.sqr(w); b = F.mul(b, z); x = F.mul(x, w); v = k;
        }
        return x;
    };
    khost::fe s = hsqrt(F.neg(three));
    put(P.fu, F.add(F.f.one, five));                        // u^3 + 5 with u = 1
    put(P.c1, F.mul(F.sub(s, F.f.one), F.inv(two)));
    put(P.s, s); put(P.c2, F.inv(three)); put(P.five, five); put(P.root, root);
    memcpy(P.t_m1_d2, &tm1d2, 32);
    dim3 grid((unsigned)((count + 127) / 128));
    if (fid == KH_FIELD_FQ) hipLaunchKernelGGL((k_srs_generate<FqParams>), grid, dim3(128), 0, C.stream, (u32)start, count, P, (uint8_t*)out_xy_dev);
    else hipLaunchKernelGGL((k_srs_generate<FpParams>), grid, dim3(128), 0, C.stream, (u32)start, count, P, (uint8_t*)out_xy_dev);
    KH_HIP(hipGetLastError());
    KH_HIP(hipStreamSynchronize(C.stream));
    return KH_OK;
}

}  // namespace kh
