// lagrange.hip -- SRS::lagrange_basis on the device (poly-commitment/src/ipa.rs:1065-1172):
// for chunk c of a domain of size n over an SRS of size s,
//     lg[c*s + j] = g[j]  (j < min((c+1)s, n) - c*s),   everything else the identity,
//     domain.ifft_in_place(&mut lg)  -- an inverse DFT over curve points --  then normalize_batch.
// Index-time work (cached per (SRS, domain) by the reference, ipa.rs:780-795), so the schedule is
// the plain one: bit-reversal load, log2(n) radix-2 decimation-in-time stages in HBM with one
// thread per butterfly (u, w^j * v) -> (u + t, u - t), a 255-bit double-and-add for each
// non-trivial twiddle, a final multiplication by n^-1, and one inversion per point to normalise.
// n/2 * log2(n) scalar multiplications: ~0.1 s at n = 2^16 on an MI355X (minutes on the CPU path).
#include "common.hpp"
#include "curve.cuh"
#include "coop.cuh"
#include "host_ec.hpp"
#include "msm.hpp"

namespace kh {

template <class BF>
__global__ void k_lag_init(const uint8_t* __restrict__ g, size_t start, size_t num_terms, unsigned log_n, uint8_t* __restrict__ A) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << log_n;
    if (i >= n) return;
    Xyzz<BF> v = Xyzz<BF>::identity();
    if (i >= start && i < start + num_terms) v = Xyzz<BF>::from_affine(Aff<BF>::load(g + (i - start) * 64));
    size_t r = log_n ? ((size_t)__brevll((unsigned long long)i) >> (64 - log_n)) : 0;
    v.store(A + r * 128);
}
// canonical (non-Montgomery) twiddles w^-e, e < n/2, from the Montgomery table
template <class SF>
__global__ void k_lag_twiddles(const u64* __restrict__ tw_mont, size_t count, u64* __restrict__ tw_plain) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    from_mont<SF>(Fe<SF>::load(tw_mont + 4 * e)).store(tw_plain + 4 * e);
}
template <class BF>
__global__ void __launch_bounds__(128)
k_lag_stage(uint8_t* __restrict__ A, const u64* __restrict__ tw_plain, unsigned log_n, unsigned log_m) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t half = (size_t)1 << (log_n - 1);
    if (t >= half) return;
    size_t m = (size_t)1 << log_m;
    size_t j = t & (m - 1), k = (t >> log_m) << (log_m + 1);
    Xyzz<BF> u = Xyzz<BF>::load(A + (k + j) * 128);
    Xyzz<BF> v = Xyzz<BF>::load(A + (k + j + m) * 128);
    if (j != 0 && !v.is_identity()) {
        u32 kw[8];
        const uint4* q = (const uint4*)(tw_plain + 4 * (j << (log_n - 1 - log_m)));
        uint4 a = q[0], b = q[1];
        kw[0] = a.x; kw[1] = a.y; kw[2] = a.z; kw[3] = a.w; kw[4] = b.x; kw[5] = b.y; kw[6] = b.z; kw[7] = b.w;
        v = scalar_mul<BF>(v, kw);
    }
    add<BF>(u, v).store(A + (k + j) * 128);
    add<BF>(u, negate<BF>(v)).store(A + (k + j + m) * 128);
}
// The same butterfly by FOUR lanes (coop.cuh): a stage is one long dependent chain per butterfly -- 255 doublings and as many additions as the twiddle has
// bits set, 23 products per bit with one lane -- on a GPU that n / 2 threads leave half empty at 2^16; with a point spread over a quad a doubling is 4
// product rounds and an addition 5: a 2.5 times shorter chain at four times the lanes.
template <class BF>
__global__ void __launch_bounds__(256)
k_lag_stage_q(uint8_t* __restrict__ A, const u64* __restrict__ tw_plain, unsigned log_n, unsigned log_m) {
    size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const size_t half = (size_t)1 << (log_n - 1);
    const bool live = t < half;
    if (!live) t = half - 1;                                 // keep whole quads in the DPP moves
    const u32 role = threadIdx.x & 3u;
    const size_t m = (size_t)1 << log_m;
    // n / 2m butterflies share the twiddle w^j: while that is at least a wave's sixteen quads, the quads of a wave take the SAME j (and sixteen
    // different k), so the ladder's additions are executed for the set bits of one twiddle only, not for the union over sixteen
    const size_t per = half >> log_m;
    const size_t j = per >= 16 ? t / per : (t & (m - 1));
    const size_t k = (per >= 16 ? t % per : (t >> log_m)) << (log_m + 1);
    const Fe<BF> u = quad_load<BF>(A + (k + j) * 128);
    Fe<BF> v = quad_load<BF>(A + (k + j + m) * 128);
    const bool v_id = quad_flag<QP_BCAST2>(v.is_zero());
    if (j != 0 && !v_id) {
        u32 kw[8];
        const uint4* q = (const uint4*)(tw_plain + 4 * (j << (log_n - 1 - log_m)));
        const uint4 a = q[0], b = q[1];
        kw[0] = a.x; kw[1] = a.y; kw[2] = a.z; kw[3] = a.w; kw[4] = b.x; kw[5] = b.y; kw[6] = b.z; kw[7] = b.w;
        v = quad_scalar_mul<BF>(v, kw);
    }
    const Fe<BF> s = quad_add<BF>(u, v);
    const Fe<BF> d = quad_add<BF>(u, role == 1 ? neg<BF>(v) : v);
    if (live) { quad_store<BF>(A + (k + j) * 128, s); quad_store<BF>(A + (k + j + m) * 128, d); }
}
// A[i] <- n^-1 * A[i], normalised to affine; identity -> zeros + flag
template <class BF>
__global__ void __launch_bounds__(128)
k_lag_finish(const uint8_t* __restrict__ A, size_t n, const u64* __restrict__ ninv_plain, uint8_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz<BF> v = Xyzz<BF>::load(A + i * 128);
    Fe<BF> x = Fe<BF>::zero(), y = Fe<BF>::zero();
    uint8_t inf = 1;
    if (!v.is_identity()) {
        u32 kw[8];
#pragma unroll
        for (int w = 0; w < 4; w++) { u64 l = ninv_plain[w]; kw[2 * w] = (u32)l; kw[2 * w + 1] = (u32)(l >> 32); }
        v = scalar_mul<BF>(v, kw);
        if (!v.is_identity()) {
            Fe<BF> izzz = inv<BF>(v.zzz);
            Fe<BF> izz = sqr<BF>(mul<BF>(izzz, v.zz));
            x = mul<BF>(v.x, izz); y = mul<BF>(v.y, izzz);
            inf = 0;
        }
    }
    x.store(out_xy + i * 64); y.store(out_xy + i * 64 + 32);
    out_inf[i] = inf;
}

template <class BF, class SF>
static int lagrange_t(Context& C, int sfield, const void* g_dev, size_t srs_size, unsigned log_n, unsigned chunk,
                      void* out_xy_dev, uint8_t* out_inf_dev) {
    const size_t n = (size_t)1 << log_n;
    const size_t start = (size_t)chunk * srs_size;
    KH_REQUIRE(start < n, "chunk %u is beyond the domain", chunk);
    const size_t num_terms = ((size_t)(chunk + 1) * srs_size < n ? (size_t)(chunk + 1) * srs_size : n) - start;
    DevBuf &A = C.scratch("lagrange_points"), &tw = C.scratch("lagrange_twiddles");
    int rc;
    if ((rc = A.reserve(n * 128))) return rc;
    const size_t half = n > 1 ? n / 2 : 1;
    if ((rc = tw.reserve(half * 64 + 64 + 32 * 32))) return rc;
    hipStream_t s = C.stream;
    // inverse-root twiddles (Montgomery) -> canonical integers
    khost::Fld F(sfield);
    khost::fe w = ntt_host_root(sfield, log_n, 1);
    // powers by repeated multiplication on the host would be O(n); reuse the device builder from ntt.hip
    u64* tw_mont = tw.as<u64>();
    u64* tw_plain = tw.as<u64>() + half * 4;
    u64* ninv_dev = tw_plain + half * 4;
    if ((rc = ntt_build_twiddles(C, sfield, log_n, 1, tw_mont))) return rc;
    hipLaunchKernelGGL((k_lag_twiddles<SF>), dim3((unsigned)((half + 255) / 256)), dim3(256), 0, s, tw_mont, half, tw_plain);
    khost::fe nn = {{(u64)n, 0, 0, 0}};
    khost::fe ninv = F.from_mont(F.inv(F.to_mont(nn)));
    (void)w;
    KH_HIP(hipMemcpyAsync(ninv_dev, &ninv, 32, hipMemcpyHostToDevice, s));
    KH_HIP(hipStreamSynchronize(s));
    hipLaunchKernelGGL((k_lag_init<BF>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint8_t*)g_dev, start, num_terms, log_n, A.as<uint8_t>());
    static const bool lag_quad = !(getenv("KH_LAG_QUAD") && atoi(getenv("KH_LAG_QUAD")) == 0);      // 0: one lane per butterfly (round 1's kernel)
    for (unsigned lm = 0; lm < log_n; lm++) {
        if (lag_quad && n >= 8)
            hipLaunchKernelGGL((k_lag_stage_q<BF>), dim3((unsigned)((4 * (n / 2) + 255) / 256)), dim3(256), 0, s, A.as<uint8_t>(), tw_plain, log_n, lm);
        else
            hipLaunchKernelGGL((k_lag_stage<BF>), dim3((unsigned)((n / 2 + 127) / 128)), dim3(128), 0, s, A.as<uint8_t>(), tw_plain, log_n, lm);
    }
    hipLaunchKernelGGL((k_lag_finish<BF>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, A.as<uint8_t>(), n, ninv_dev, (uint8_t*)out_xy_dev, out_inf_dev);
    KH_HIP(hipGetLastError());
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}

int lagrange_run(Context& C, int curve, const void* g_dev, size_t srs_size, unsigned log_n, unsigned chunk, void* out_xy_dev, uint8_t* out_inf_dev) {
    if (curve == KH_CURVE_VESTA) return lagrange_t<FqParams, FpParams>(C, KH_FIELD_FP, g_dev, srs_size, log_n, chunk, out_xy_dev, out_inf_dev);
    return lagrange_t<FpParams, FqParams>(C, KH_FIELD_FQ, g_dev, srs_size, log_n, chunk, out_xy_dev, out_inf_dev);
}

}  // namespace kh
