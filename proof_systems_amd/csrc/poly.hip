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
#include "host_ec.hpp"

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
// ---------------------------------------------------------------- chunked evaluation
// kh_evaluate_chunks(_batch)_dev: the prover's zeta / zeta*omega evaluations (prover.rs:939-1010, PolyComm-chunked).  A chunk is cut
// into segments; one 256-thread workgroup per (segment, point): thread t runs Horner in y = x^256 down its residue class of the
// segment, multiplies by x^(segment start + t) -- the product of the host-supplied powers x^(2^k) over the set bits -- and a block
// tree adds the 256 values.  The host adds the segment values.  A proof's 50 chunks x 2 points used to be 100 workgroups of 1024
// threads on 100 CUs, each SIMD issuing four 94-product chains (171 us); 800 smaller workgroups spread the same products over the
// whole chip, and the lone ft polynomial (one chunk, one point: the same 171 us on ONE CU) becomes 32 short segments.
struct ChunkDesc { const u64* ptr; u64 len; u64 out; u64 start; };   // a segment: `len` coefficients from ptr, value slot `out`, x^start factor
static constexpr int EVAL_T = 256, EVAL_PW = 40;                     // powers x^(2^k), k < EVAL_PW, per point
template <class F>
__global__ void __launch_bounds__(EVAL_T)
k_eval_chunks(const ChunkDesc* __restrict__ tab, const u64* __restrict__ pw, size_t nseg, u64* __restrict__ out) {
    __shared__ u32 sh[EVAL_T * 8];
    const ChunkDesc d = tab[blockIdx.x];
    const size_t p = blockIdx.y;
    const size_t len = d.len;
    const u64* xp = pw + 4 * EVAL_PW * p;
    const u32 t = threadIdx.x;
    Fe<F> acc = Fe<F>::zero();
    if (t < len) {
        const Fe<F> y = Fe<F>::load(xp + 4 * 8);             // x^256
        size_t last = t + ((len - 1 - t) / EVAL_T) * EVAL_T; // highest index of this thread's residue class
        for (size_t j = last;; j -= EVAL_T) {
            acc = add<F>(mul<F>(acc, y), Fe<F>::load(d.ptr + 4 * j));
            if (j < EVAL_T) break;
        }
        u64 e = d.start + t;
        for (int k = 0; e; k++, e >>= 1)
            if (e & 1) acc = mul<F>(acc, Fe<F>::load(xp + 4 * k));
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k * EVAL_T + t] = acc.v[k];
    __syncthreads();
    for (int s = EVAL_T / 2; s >= 1; s >>= 1) {
        if ((int)t < s) {
            Fe<F> o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[k * EVAL_T + t + s];
            acc = add<F>(acc, o);
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k * EVAL_T + t] = acc.v[k];
        }
        __syncthreads();
    }
    if (t == 0) acc.store(out + 4 * (d.out + p * nseg));
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

// ---------------------------------------------------------------- scans over field elements
// Inclusive prefix (or suffix, `rev`) scan under + or *: the building block of ark_ff::batch_inversion
// (permutation.rs:533), of the running product z of perm_aggreg (permutation.rs:556-563) and of division by a
// linear factor (permutation.rs:300-327).  Three launches: 2048-element tiles (8 per thread, sequential, then a
// Hillis-Steele pass over the 256 thread totals in LDS), a single-block scan of the tile totals, the fix-up.
static constexpr int SCAN_E = 8, SCAN_T = 256, SCAN_TILE = SCAN_E * SCAN_T;
template <class F, int OP> __device__ __forceinline__ Fe<F> scan_op(const Fe<F>& a, const Fe<F>& b) { return OP == 0 ? add<F>(a, b) : mul<F>(a, b); }
template <class F, int OP> __device__ __forceinline__ Fe<F> scan_id() { return OP == 0 ? Fe<F>::zero() : Fe<F>::one(); }

// scans the SCAN_T values held one per thread (inclusive); returns this thread's exclusive prefix
template <class F, int OP>
__device__ __forceinline__ Fe<F> block_exclusive(Fe<F> v, u32* sh, Fe<F>* total) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k * SCAN_T + t] = v.v[k];
    __syncthreads();
    for (int d = 1; d < SCAN_T; d <<= 1) {
        Fe<F> o = scan_id<F, OP>();
        if (t >= d) {
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[k * SCAN_T + t - d];
        }
        __syncthreads();
        if (t >= d) {
            v = scan_op<F, OP>(o, v);
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k * SCAN_T + t] = v.v[k];
        }
        __syncthreads();
    }
    Fe<F> ex = scan_id<F, OP>();
    if (t > 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) ex.v[k] = sh[k * SCAN_T + t - 1];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) total->v[k] = sh[k * SCAN_T + SCAN_T - 1];
    __syncthreads();
    return ex;
}
template <class F, int OP>
__global__ void __launch_bounds__(SCAN_T)
k_scan_tiles(u64* __restrict__ data, size_t n, int rev, u64* __restrict__ totals) {
    __shared__ u32 sh[8 * SCAN_T];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_E;
    Fe<F> loc[SCAN_E];
    Fe<F> run = scan_id<F, OP>();
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        const size_t i = base + e;
        loc[e] = i < n ? Fe<F>::load(data + 4 * (rev ? n - 1 - i : i)) : scan_id<F, OP>();
        run = scan_op<F, OP>(run, loc[e]);
        loc[e] = run;
    }
    Fe<F> total;
    const Fe<F> ex = block_exclusive<F, OP>(run, sh, &total);
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        const size_t i = base + e;
        if (i < n) scan_op<F, OP>(ex, loc[e]).store(data + 4 * (rev ? n - 1 - i : i));
    }
    if (threadIdx.x == 0) total.store(totals + 4 * blockIdx.x);
}
// exclusive scan of the tile totals in place (one block; <= SCAN_TILE tiles = 4M elements)
template <class F, int OP>
__global__ void __launch_bounds__(SCAN_T)
k_scan_totals(u64* __restrict__ totals, size_t ntiles) {
    __shared__ u32 sh[8 * SCAN_T];
    const size_t base = (size_t)threadIdx.x * SCAN_E;
    Fe<F> loc[SCAN_E];
    Fe<F> run = scan_id<F, OP>();
#pragma unroll
    for (int e = 0; e < SCAN_E; e++) {
        loc[e] = run;                                       // exclusive
        if (base + e < ntiles) run = scan_op<F, OP>(run, Fe<F>::load(totals + 4 * (base + e)));
    }
    Fe<F> total;
    const Fe<F> ex = block_exclusive<F, OP>(run, sh, &total);
#pragma unroll
    for (int e = 0; e < SCAN_E; e++)
        if (base + e < ntiles) scan_op<F, OP>(ex, loc[e]).store(totals + 4 * (base + e));
}
template <class F, int OP>
__global__ void k_scan_fix(u64* __restrict__ data, size_t n, int rev, const u64* __restrict__ totals) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i < SCAN_TILE) return;
    const size_t pos = rev ? n - 1 - i : i;
    scan_op<F, OP>(Fe<F>::load(totals + 4 * (i / SCAN_TILE)), Fe<F>::load(data + 4 * pos)).store(data + 4 * pos);
}
// elementwise helpers of batch inversion and division by (x - a)
template <class F>
__global__ void k_zero_to_one(const u64* __restrict__ v, size_t n, u64* __restrict__ out) {       // zeros are skipped by batch_inversion
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> x = Fe<F>::load(v + 4 * i);
    (x.is_zero() ? Fe<F>::one() : x).store(out + 4 * i);
}
// v_i^-1 = inv_total * prefix_{i-1} * suffix_{i+1}
template <class F>
__global__ void k_batch_inv_finish(u64* __restrict__ v, const u64* __restrict__ pre, const u64* __restrict__ suf, Fe4p inv_total, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (Fe<F>::load(v + 4 * i).is_zero()) return;
    Fe<F> r = Fe<F>::load(inv_total.l);
    if (i > 0) r = mul<F>(r, Fe<F>::load(pre + 4 * (i - 1)));
    if (i + 1 < n) r = mul<F>(r, Fe<F>::load(suf + 4 * (i + 1)));
    r.store(v + 4 * i);
}
// t_k = c_k a^k
template <class F>
__global__ void k_scale_powers(const u64* __restrict__ c, Fe4p a, size_t n, size_t shift, u64* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    mul<F>(Fe<F>::load(c + 4 * k), pow_u64<F>(Fe<F>::load(a.l), k + shift)).store(out + 4 * k);
}
// out[r][j] = in[r][j] * a^j: the coefficient twist of a coset evaluation, f(a w^i) = NTT_n(c_j a^j)[i]
template <class F>
__global__ void k_coset_scale(const u64* __restrict__ in, Fe4p a, size_t n, size_t total, u64* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    mul<F>(Fe<F>::load(in + 4 * k), pow_u64<F>(Fe<F>::load(a.l), k % n)).store(out + 4 * k);
}
// q_i = S_{i+1} a^-(i+1), i < n - 1, from the suffix sums S of c_k a^k
template <class F>
__global__ void k_linear_quotient(const u64* __restrict__ S, Fe4p a_inv, size_t n, u64* __restrict__ q) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    mul<F>(Fe<F>::load(S + 4 * (i + 1)), pow_u64<F>(Fe<F>::load(a_inv.l), i + 1)).store(q + 4 * i);
}

#define KH_FIELD_DISPATCH(KERNEL, grid, block, stream, ...)                                             \
    do {                                                                                                \
        if (field == KH_FIELD_FP) hipLaunchKernelGGL((KERNEL<FpParams>), grid, block, 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<FqParams>), grid, block, 0, stream, __VA_ARGS__);                \
        KH_HIP(hipGetLastError());                                                                      \
    } while (0)

#define g_poly_tab (kh::ctx().scratch("poly_tab"))

int poly_lincomb(Context& C, int field, const uint64_t* const* segs_dev, const size_t* lens, const uint64_t* scales, size_t m,
                 uint64_t* out_dev, size_t out_len) {
    if (out_len == 0) return KH_OK;
    int rc;
    if ((rc = g_poly_tab.reserve(m * (8 + 8 + 32) + 64))) return rc;
    char* tab = g_poly_tab.as<char>();
    std::vector<u64> lens64(lens, lens + m);
    hipStream_t s = C.stream;
    if (m && (rc = C.stage_upload(tab, {{segs_dev, m * 8}, {lens64.data(), m * 8}, {scales, m * 32}}))) return rc;
    KH_FIELD_DISPATCH(k_lincomb, dim3((unsigned)((out_len + 255) / 256)), dim3(256), s,
                      (const u64* const*)tab, (const u64*)(tab + m * 8), (const u64*)(tab + m * 16), m, out_len, out_dev);
    return KH_OK;
}
int poly_b_init(Context& C, int field, const uint64_t* elm, const uint64_t* scales, size_t k, size_t n, uint64_t* out_dev) {
    if (n == 0) return KH_OK;
    int rc;
    if ((rc = g_poly_tab.reserve(k * 64 + 64))) return rc;
    hipStream_t s = C.stream;
    if (k && (rc = C.stage_upload(g_poly_tab.p, {{elm, k * 32}, {scales, k * 32}}))) return rc;
    KH_FIELD_DISPATCH(k_b_init, dim3((unsigned)((n + 255) / 256)), dim3(256), s,
                      g_poly_tab.as<u64>(), g_poly_tab.as<u64>() + 4 * k, k, n, out_dev);
    return KH_OK;
}
// polynomial j (polys_dev[j], lens[j] coefficients) is cut into num_chunks[j] chunks of `chunk`; out (host) receives, per
// polynomial in order, npts x num_chunks[j] values: out_j[p][c] = chunk_c(points[p])
int poly_eval_chunks(Context& C, int field, const uint64_t* const* polys_dev, const size_t* lens, const size_t* num_chunks, size_t m, size_t chunk,
                     const uint64_t* points, size_t npts, uint64_t* out) {
    size_t nchunks = 0;
    for (size_t j = 0; j < m; j++) nchunks += num_chunks[j];
    if (nchunks == 0 || npts == 0) return KH_OK;
    // segment length: long enough to amortise the x^(start + t) product, short enough to fill the chip
    const size_t seg = nchunks * npts >= 32 ? 8192 : 2048;
    struct Slot { size_t first, count; };          // segments of one (polynomial, chunk)
    std::vector<ChunkDesc> tab; std::vector<Slot> slots;
    for (size_t j = 0; j < m; j++)
        for (size_t c = 0; c < num_chunks[j]; c++) {
            const size_t off = c * chunk;
            const size_t len = off >= lens[j] ? 0 : std::min(chunk, lens[j] - off);
            slots.push_back(Slot{tab.size(), 0});
            for (size_t b = 0; b < len; b += seg) {
                tab.push_back(ChunkDesc{polys_dev[j] + 4 * (off + b), (u64)std::min(seg, len - b), (u64)tab.size(), (u64)b});
                slots.back().count++;
            }
        }
    const size_t nseg = tab.size(), base = nchunks * npts;
    khost::Fld F(field);
    if (nseg == 0) { memset(out, 0, base * 32); return KH_OK; }
    std::vector<khost::fe> pw(npts * EVAL_PW);
    for (size_t p = 0; p < npts; p++) {
        memcpy(&pw[p * EVAL_PW], points + 4 * p, 32);
        for (int k = 1; k < EVAL_PW; k++) pw[p * EVAL_PW + k] = F.sqr(pw[p * EVAL_PW + k - 1]);
    }
    int rc;
    const size_t tab_bytes = nseg * sizeof(ChunkDesc), pw_bytes = pw.size() * 32, res_bytes = nseg * npts * 32;
    if ((rc = g_poly_tab.reserve(tab_bytes + pw_bytes + res_bytes + 64))) return rc;
    hipStream_t s = C.stream;
    char* d_tab = g_poly_tab.as<char>(); u64* d_pw = (u64*)(d_tab + tab_bytes); u64* res = d_pw + pw_bytes / 8;
    std::vector<khost::fe> part(nseg * npts);
    if ((rc = C.stage_upload(d_tab, {{tab.data(), tab_bytes}, {pw.data(), pw_bytes}}))) return rc;
    KH_FIELD_DISPATCH(k_eval_chunks, dim3((unsigned)nseg, (unsigned)npts), dim3(EVAL_T), s, (const ChunkDesc*)d_tab, (const u64*)d_pw, nseg, res);
    KH_HIP(hipMemcpyAsync(part.data(), res, res_bytes, hipMemcpyDeviceToHost, s));
    KH_HIP(hipStreamSynchronize(s));
    // out_j[p][c] for polynomial j in order: npts x num_chunks[j]
    size_t o = 0, sl = 0;
    for (size_t j = 0; j < m; j++) {
        for (size_t p = 0; p < npts; p++)
            for (size_t c = 0; c < num_chunks[j]; c++) {
                const Slot& S = slots[sl + c];
                khost::fe v; memset(&v, 0, 32);
                for (size_t q = 0; q < S.count; q++) v = F.add(v, part[p * nseg + S.first + q]);
                memcpy(out + 4 * (o + p * num_chunks[j] + c), &v, 32);
            }
        o += npts * num_chunks[j]; sl += num_chunks[j];
    }
    return KH_OK;
}
int poly_div_vanishing(Context& C, int field, const uint64_t* f_dev, size_t len, size_t n, uint64_t* q_dev, uint64_t* r_dev) {
    hipStream_t s = C.stream;
    KH_FIELD_DISPATCH(k_div_vanishing, dim3((unsigned)((n + 255) / 256)), dim3(256), s, f_dev, len, n, q_dev, r_dev);
    return KH_OK;
}

#define KH_SCAN_DISPATCH(KERNEL, grid, block, stream, ...)                                                              \
    do {                                                                                                                 \
        if (field == KH_FIELD_FP) { if (op == 0) hipLaunchKernelGGL((KERNEL<FpParams, 0>), grid, block, 0, stream, __VA_ARGS__); \
                                    else hipLaunchKernelGGL((KERNEL<FpParams, 1>), grid, block, 0, stream, __VA_ARGS__); }          \
        else { if (op == 0) hipLaunchKernelGGL((KERNEL<FqParams, 0>), grid, block, 0, stream, __VA_ARGS__);                     \
               else hipLaunchKernelGGL((KERNEL<FqParams, 1>), grid, block, 0, stream, __VA_ARGS__); }                           \
        KH_HIP(hipGetLastError());                                                                                       \
    } while (0)

#define g_scan_tot (kh::ctx().scratch("scan_tot"))
#define g_scan_a (kh::ctx().scratch("scan_a"))
#define g_scan_b (kh::ctx().scratch("scan_b"))

static int scan_enqueue(hipStream_t s, int field, int op, int rev, uint64_t* data_dev, size_t n) {
    if (n == 0) return KH_OK;
    const size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    KH_REQUIRE(ntiles <= (size_t)SCAN_TILE, "scan of %zu elements: more than %d tiles", n, SCAN_TILE);
    int rc;
    if ((rc = g_scan_tot.reserve(ntiles * 32))) return rc;
    KH_SCAN_DISPATCH(k_scan_tiles, dim3((unsigned)ntiles), dim3(SCAN_T), s, data_dev, n, rev, g_scan_tot.as<u64>());
    if (ntiles > 1) {
        KH_SCAN_DISPATCH(k_scan_totals, dim3(1), dim3(SCAN_T), s, g_scan_tot.as<u64>(), ntiles);
        KH_SCAN_DISPATCH(k_scan_fix, dim3((unsigned)((n + 255) / 256)), dim3(256), s, data_dev, n, rev, (const u64*)g_scan_tot.as<u64>());
    }
    return KH_OK;
}
// evaluations of `batch` polynomials of n = 2^log2_n coefficients on the coset shift * <w_n> (natural order); asynchronous
// on the main stream like kh_ntt_dev
int poly_coset_ntt(Context& C, int field, const uint64_t* coeffs_dev, unsigned log2_n, const uint64_t shift[4], uint64_t* out_dev, size_t batch) {
    const size_t n = (size_t)1 << log2_n, total = n * batch;
    if (total == 0) return KH_OK;
    Fe4p a; memcpy(a.l, shift, 32);
    hipStream_t s = C.stream;
    KH_FIELD_DISPATCH(k_coset_scale, dim3((unsigned)((total + 255) / 256)), dim3(256), s, coeffs_dev, a, n, total, out_dev);
    return ntt_run(C, field, out_dev, log2_n, 0, batch);
}
int poly_scan(Context& C, int field, int op, int rev, uint64_t* data_dev, size_t n) {
    return scan_enqueue(C.stream, field, op, rev, data_dev, n);
}
// ark_ff::batch_inversion: every non-zero element replaced by its inverse, zeros untouched
int poly_batch_inversion(Context& C, int field, uint64_t* v_dev, size_t n) {
    if (n == 0) return KH_OK;
    int rc;
    if ((rc = g_scan_a.reserve(n * 32))) return rc;
    if ((rc = g_scan_b.reserve(n * 32))) return rc;
    hipStream_t s = C.stream;
    dim3 g((unsigned)((n + 255) / 256));
    KH_FIELD_DISPATCH(k_zero_to_one, g, dim3(256), s, (const u64*)v_dev, n, g_scan_a.as<u64>());
    KH_HIP(hipMemcpyAsync(g_scan_b.p, g_scan_a.p, n * 32, hipMemcpyDeviceToDevice, s));
    if ((rc = scan_enqueue(s, field, 1, 0, g_scan_a.as<u64>(), n))) return rc;        // prefix products
    if ((rc = scan_enqueue(s, field, 1, 1, g_scan_b.as<u64>(), n))) return rc;        // suffix products
    khost::fe total;
    KH_HIP(hipMemcpyAsync(&total, g_scan_a.as<char>() + (n - 1) * 32, 32, hipMemcpyDeviceToHost, s));
    KH_HIP(hipStreamSynchronize(s));
    Fe4p inv; host_field_inverse(field, total.l, inv.l);   // one inversion, on the host (20 us against ~0.2 ms for a lone GPU thread)
    KH_FIELD_DISPATCH(k_batch_inv_finish, g, dim3(256), s, v_dev, (const u64*)g_scan_a.as<u64>(), (const u64*)g_scan_b.as<u64>(), inv, n);
    return KH_OK;
}
// f = q (x - a) + rem, rem = f(a): q_i = a^-(i+1) sum_{k > i} c_k a^k
// rem: the remainder to the HOST (synchronises the stream), or -- rem == nullptr -- rem_dev: four limbs on the device, nothing waits
int poly_divide_by_linear(Context& C, int field, const uint64_t* f_dev, size_t len, const uint64_t a[4], uint64_t* q_dev, uint64_t rem[4], uint64_t* rem_dev) {
    hipStream_t s = C.stream;
    if (len == 0) { if (rem) memset(rem, 0, 32); else if (rem_dev) KH_HIP(hipMemsetAsync(rem_dev, 0, 32, s)); return KH_OK; }
    if ((a[0] | a[1] | a[2] | a[3]) == 0) {                 // division by x: a shift
        if (len > 1) KH_HIP(hipMemcpyAsync(q_dev, f_dev + 4, (len - 1) * 32, hipMemcpyDeviceToDevice, s));
        if (rem) { KH_HIP(hipMemcpyAsync(rem, f_dev, 32, hipMemcpyDeviceToHost, s)); KH_HIP(hipStreamSynchronize(s)); }
        else if (rem_dev) KH_HIP(hipMemcpyAsync(rem_dev, f_dev, 32, hipMemcpyDeviceToDevice, s));
        return KH_OK;
    }
    int rc;
    if ((rc = g_scan_a.reserve(len * 32))) return rc;
    Fe4p av, ai; memcpy(av.l, a, 32); host_field_inverse(field, a, ai.l);
    dim3 g((unsigned)((len + 255) / 256));
    KH_FIELD_DISPATCH(k_scale_powers, g, dim3(256), s, f_dev, av, len, (size_t)0, g_scan_a.as<u64>());
    if ((rc = scan_enqueue(s, field, 0, 1, g_scan_a.as<u64>(), len))) return rc;      // suffix sums
    if (len > 1) KH_FIELD_DISPATCH(k_linear_quotient, g, dim3(256), s, (const u64*)g_scan_a.as<u64>(), ai, len, q_dev);
    if (rem) { KH_HIP(hipMemcpyAsync(rem, g_scan_a.p, 32, hipMemcpyDeviceToHost, s)); KH_HIP(hipStreamSynchronize(s)); }
    else if (rem_dev) KH_HIP(hipMemcpyAsync(rem_dev, g_scan_a.p, 32, hipMemcpyDeviceToDevice, s));
    return KH_OK;
}

// Deferred invariant checks: *flags |= 1 << bit when any of the n elements of v differs from `expect` (as 4 x u64 words; no field arithmetic)
__global__ void k_check_equal(const u64* __restrict__ v, size_t nwords, Fe4p expect, u32* __restrict__ flags, u32 bit) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < nwords; i += stride) bad |= v[i] != expect.l[i & 3];
    if (__any(bad) && (threadIdx.x & 63u) == 0) atomicOr(flags, 1u << bit);
}
int poly_check_equal(Context& C, const uint64_t* v_dev, size_t n, const uint64_t* expect, uint32_t* flags_dev, unsigned bit) {
    if (n == 0) return KH_OK;
    Fe4p e; if (expect) memcpy(e.l, expect, 32); else memset(e.l, 0, 32);
    const size_t nwords = 4 * n;
    unsigned blocks = (unsigned)std::min<size_t>((nwords + 255) / 256, 1024);
    hipLaunchKernelGGL(k_check_equal, dim3(blocks), dim3(256), 0, C.stream, v_dev, nwords, e, flags_dev, (u32)bit);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

}  // namespace kh
