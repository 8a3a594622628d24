// poly.hip -- the coefficient-vector operations that sit between the transforms / commitments and the opening
// proof in ProverProof::create, on device-resident vectors (the callers either side of the hot path):
//   combine_polys            p = sum_i polyscale^i * chunk_i              poly-commitment/src/utils.rs:103-206
//   b_init                   b[j] = sum_i evalscale^i * elm_i^j           poly-commitment/src/ipa.rs:863-888
//   evaluate_chunks          chunk_c(x) for every chunk of a polynomial   utils/src/chunked_polynomial.rs:21-28
//   divide_by_vanishing_poly f = q (x^n - 1) + r                          kimchi/src/prover.rs:903 (ark-poly)
// All four are element-wise or short reductions over 32-byte Montgomery elements; none is a bottleneck
// (a few field products per element), the point is that their inputs and outputs never leave HBM.
#include "common.hpp"
#include "field.cuh"
#include "msm.hpp"

namespace kh {

struct Fe4p { u64 l[4]; };

// out[i] = sum_j scale[j] * seg_j[i]  (i < len_j); segment table on the device
template <class F>
__global__ void k_lincomb(const u64* const* __restrict__ segs, const u64* __restrict__ lens, const u64* __restrict__ scales, size_t m,
                          size_t out_len, u64* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_len) return;
    Fe<F> acc = Fe<F>::zero();
    for (size_t j = 0; j < m; j++)
        if (i < lens[j]) acc = add<F>(acc, mul<F>(Fe<F>::load(scales + 4 * j), Fe<F>::load(segs[j] + 4 * i)));
    acc.store(out + 4 * i);
}
template <class F>
__device__ __forceinline__ Fe<F> pow_u64(Fe<F> base, u64 e) {
    Fe<F> acc = Fe<F>::one();
    while (e) { if (e & 1) acc = mul<F>(acc, base); e >>= 1; if (e) base = sqr<F>(base); }
    return acc;
}
// b[j] = sum_i scale_i * elm_i^j, scale_i = evalscale^i (precomputed on the host)
template <class F>
__global__ void k_b_init(const u64* __restrict__ elm, const u64* __restrict__ scales, size_t k, size_t n, u64* __restrict__ out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Fe<F> acc = Fe<F>::zero();
    for (size_t i = 0; i < k; i++) acc = add<F>(acc, mul<F>(Fe<F>::load(scales + 4 * i), pow_u64<F>(Fe<F>::load(elm + 4 * i), j)));
    acc.store(out + 4 * j);
}
// one block per (chunk, point): sum_{j < len} c[j] x^j with thread t running Horner over j = t, t+T, ... in y = x^T
template <class F>
__global__ void __launch_bounds__(256)
k_eval_chunks(const u64* __restrict__ coeffs, size_t total_len, size_t chunk, const u64* __restrict__ points, u64* __restrict__ out) {
    __shared__ u32 sh[256 * 8];
    const size_t c = blockIdx.x, p = blockIdx.y;
    const size_t base = c * chunk;
    const size_t len = base >= total_len ? 0 : (total_len - base < chunk ? total_len - base : chunk);
    const Fe<F> x = Fe<F>::load(points + 4 * p);
    const u32 T = blockDim.x, t = threadIdx.x;
    Fe<F> acc = Fe<F>::zero();
    if (t < len) {
        const Fe<F> y = pow_u64<F>(x, T);
        size_t last = t + ((len - 1 - t) / T) * T;           // highest index of this thread's residue class
        for (size_t j = last;; j -= T) {
            acc = add<F>(mul<F>(acc, y), Fe<F>::load(coeffs + 4 * (base + j)));
            if (j < T) break;
        }
        acc = mul<F>(acc, pow_u64<F>(x, t));
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k * 256 + t] = acc.v[k];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)t < s) {
            Fe<F> o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[k * 256 + t + s];
            acc = add<F>(acc, o);
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k * 256 + t] = acc.v[k];
        }
        __syncthreads();
    }
    if (t == 0) acc.store(out + 4 * (p * gridDim.x + c));
}
// f = q (x^n - 1) + r:  q[i] = sum_{k >= 1} f[i + k n],  r[i] = sum_{k >= 0} f[i + k n] (i < n).
// Thread per residue i: suffix sums down the class.
template <class F>
__global__ void k_div_vanishing(const u64* __restrict__ f, size_t len, size_t n, u64* __restrict__ q, u64* __restrict__ r) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> acc = Fe<F>::zero();
    if (i < len) {
        size_t top = i + ((len - 1 - i) / n) * n;
        for (size_t j = top; j >= n + i; j -= n) {              // j = i + k n, k >= 1: q[j - n] = sum of f at j, j + n, ...
            acc = add<F>(acc, Fe<F>::load(f + 4 * j));
            acc.store(q + 4 * (j - n));
        }
        acc = add<F>(acc, Fe<F>::load(f + 4 * i));
    }
    acc.store(r + 4 * i);
}

#define KH_FIELD_DISPATCH(KERNEL, grid, block, stream, ...)                                             \
    do {                                                                                                \
        if (field == KH_FIELD_FP) hipLaunchKernelGGL((KERNEL<FpParams>), grid, block, 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<FqParams>), grid, block, 0, stream, __VA_ARGS__);                \
        KH_HIP(hipGetLastError());                                                                      \
    } while (0)

static DevBuf g_poly_tab;

int poly_lincomb(Context& C, int field, const uint64_t* const* segs_dev, const size_t* lens, const uint64_t* scales, size_t m,
                 uint64_t* out_dev, size_t out_len) {
    if (out_len == 0) return KH_OK;
    int rc;
    if ((rc = g_poly_tab.reserve(m * (8 + 8 + 32) + 64))) return rc;
    char* tab = g_poly_tab.as<char>();
    std::vector<u64> lens64(lens, lens + m);
    hipStream_t s = C.stream;
    if (m) {
        KH_HIP(hipMemcpyAsync(tab, segs_dev, m * 8, hipMemcpyHostToDevice, s));
        KH_HIP(hipMemcpyAsync(tab + m * 8, lens64.data(), m * 8, hipMemcpyHostToDevice, s));
        KH_HIP(hipMemcpyAsync(tab + m * 16, scales, m * 32, hipMemcpyHostToDevice, s));
    }
    KH_FIELD_DISPATCH(k_lincomb, dim3((unsigned)((out_len + 255) / 256)), dim3(256), s,
                      (const u64* const*)tab, (const u64*)(tab + m * 8), (const u64*)(tab + m * 16), m, out_len, out_dev);
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}
int poly_b_init(Context& C, int field, const uint64_t* elm, const uint64_t* scales, size_t k, size_t n, uint64_t* out_dev) {
    if (n == 0) return KH_OK;
    int rc;
    if ((rc = g_poly_tab.reserve(k * 64 + 64))) return rc;
    hipStream_t s = C.stream;
    if (k) {
        KH_HIP(hipMemcpyAsync(g_poly_tab.p, elm, k * 32, hipMemcpyHostToDevice, s));
        KH_HIP(hipMemcpyAsync(g_poly_tab.as<char>() + k * 32, scales, k * 32, hipMemcpyHostToDevice, s));
    }
    KH_FIELD_DISPATCH(k_b_init, dim3((unsigned)((n + 255) / 256)), dim3(256), s,
                      g_poly_tab.as<u64>(), g_poly_tab.as<u64>() + 4 * k, k, n, out_dev);
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}
int poly_eval_chunks(Context& C, int field, const uint64_t* coeffs_dev, size_t len, size_t chunk, size_t num_chunks,
                     const uint64_t* points, size_t npts, uint64_t* out) {
    if (num_chunks == 0 || npts == 0) return KH_OK;
    int rc;
    if ((rc = g_poly_tab.reserve(npts * 32 + num_chunks * npts * 32 + 64))) return rc;
    hipStream_t s = C.stream;
    u64* pts = g_poly_tab.as<u64>(); u64* res = pts + 4 * npts;
    KH_HIP(hipMemcpyAsync(pts, points, npts * 32, hipMemcpyHostToDevice, s));
    KH_FIELD_DISPATCH(k_eval_chunks, dim3((unsigned)num_chunks, (unsigned)npts), dim3(256), s, coeffs_dev, len, chunk, (const u64*)pts, res);
    KH_HIP(hipMemcpyAsync(out, res, num_chunks * npts * 32, hipMemcpyDeviceToHost, s));
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}
int poly_div_vanishing(Context& C, int field, const uint64_t* f_dev, size_t len, size_t n, uint64_t* q_dev, uint64_t* r_dev) {
    hipStream_t s = C.stream;
    KH_FIELD_DISPATCH(k_div_vanishing, dim3((unsigned)((n + 255) / 256)), dim3(256), s, f_dev, len, n, q_dev, r_dev);
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}

}  // namespace kh
