// msm.hip -- bucket-method (Pippenger) multi-scalar multiplication on gfx950.
//
// Replaces ark_ec::VariableBaseMSM::{msm, msm_bigint} at the reference call sites
// poly-commitment/src/ipa.rs:649,658-659,672 and commitment.rs:382 (see include/kimchi_hip.h).
// The result of an MSM is a unique group element, so only the final affine point is
// compared with the reference; the schedule below is designed for the MI355X, not
// translated from ark-ec:
//
//   1 digits     thread/scalar: Montgomery -> canonical (reduction half of a mont-mul),
//                signed c-bit digits d in (-2^(c-1), 2^(c-1)]  ->  int32 digit matrix [k][W][n]
//   2 histogram  block (slice, window, msm): bucket histogram of its slice in LDS
//                (skew-proof: the bench circuit puts n-10 of n scalars in ONE bucket;
//                LDS same-address atomics cost cycles, not a serialised HBM atomic)
//   3 scan       per-key totals + exclusive scans -> bucket offsets (counting sort, no
//                global atomics, deterministic offsets)
//   4 scatter    block (slice, window, msm): LDS cursor per bucket, writes point indices
//                (sign in bit 31) grouped by bucket
//   5 accumulate thread/task, a task = <= K consecutive entries of ONE bucket:
//                XYZZ accumulator + mixed additions of gathered affine points (8M+2S).
//                Large buckets are split into many tasks, so skew costs nothing here.
//   6 bucket sum thread/bucket adds its (few) task partials; buckets with many partials
//                go to a wave-per-bucket tree (shuffles)
//   7 reduce     sum_b (b+1) * B_b per window: thread/segment running sums + small
//                double-and-add for the segment offset, then a block tree per window
//   8 finish     host: Horner over the W window sums, XYZZ -> affine (1 inversion)
//
// With precomputed window multiples 2^(cw) * P_i (static bases; uses the 288 GB of HBM)
// all windows share one bucket set and step 8's Horner disappears.
#include <hip/hip_ext.h>
#include <math.h>
#include "common.hpp"
#include "curve.cuh"
#include "coop.cuh"
#include "field29.cuh"
#include "host_ec.hpp"
#include "msm.hpp"
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

namespace kh {

// Every kernel of the pipeline except the accumulation raises its wave priority: VALU issue on a SIMD is arbitrated by priority, then
// age, so the short sort / tail kernels of the neighbouring jobs -- YOUNGER than the four resident accumulation waves they run
// underneath -- otherwise get only the leftover issue slots (a 24 us k_digits took 511 us, a 65 us bucket sum 800 us, and the
// four pipeline slots became latency-bound: tools/timeline.py).  They are a few percent of the work; the accumulation barely notices.
#define KH_HIGH_PRIO() __builtin_amdgcn_s_setprio(3)

// One persistent helper thread for the host part of msm_finish: on the plain (per-window) path every
// result needs a ~256-doubling Horner fold (~70 us of CPU); the L and R commitments of an IPA round
// are finished side by side instead of one after the other.
class HostHelper {
    std::thread th_; std::mutex m_; std::condition_variable cv_, done_cv_;
    std::function<void()> job_; bool has_job_ = false, busy_ = false, stop_ = false;
    void loop() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [&] { return has_job_ || stop_; });
            if (stop_) return;
            std::function<void()> j = std::move(job_); has_job_ = false; busy_ = true;
            lk.unlock(); j(); lk.lock();
            busy_ = false; done_cv_.notify_all();
        }
    }
  public:
    HostHelper() : th_([this] { loop(); }) {}
    ~HostHelper() { { std::lock_guard<std::mutex> lk(m_); stop_ = true; } cv_.notify_all(); th_.join(); }
    void run(std::function<void()> j) { { std::lock_guard<std::mutex> lk(m_); job_ = std::move(j); has_job_ = true; } cv_.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(m_); done_cv_.wait(lk, [&] { return !has_job_ && !busy_; }); }
};
static HostHelper& host_helper(Context* c) {            // one per context (msm_finish runs under that context's lock; a helper takes one job at a time)
    static std::map<Context*, std::unique_ptr<HostHelper>> h; static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    auto& p = h[c];
    if (!p) p.reset(new HostHelper);
    return *p;
}

// ------------------------------------------------------------------------------------ scan
static constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_B = SCAN_T * SCAN_I;

__global__ void k_scan_block(const u32* in, u32* out, u32* sums, size_t n) {
    KH_HIGH_PRIO();
    __shared__ u32 sh[SCAN_T];
    size_t base = (size_t)blockIdx.x * SCAN_B + (size_t)threadIdx.x * SCAN_I;
    u32 v[SCAN_I]; u32 tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) { v[i] = (base + i < n) ? in[base + i] : 0u; tot += v[i]; }
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < SCAN_T; d <<= 1) {     // Hillis-Steele inclusive scan of the thread totals
        u32 t = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    u32 excl = sh[threadIdx.x] - tot;
    if (threadIdx.x == SCAN_T - 1 && sums) sums[blockIdx.x] = sh[threadIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) { if (base + i < n) out[base + i] = excl; excl += v[i]; }
}
__global__ void k_scan_add(u32* out, const u32* sums, size_t n) {
    KH_HIGH_PRIO();
    size_t i = (size_t)blockIdx.x * SCAN_B + threadIdx.x;
    u32 add = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_I; k++, i += SCAN_T) if (i < n) out[i] += add;
}
int exclusive_scan_u32(const u32* in, u32* out, size_t n, DevBuf& tmp, hipStream_t s) {
    if (n == 0) return KH_OK;
    // level sizes
    std::vector<size_t> lv; lv.push_back(n);
    while (lv.back() > (size_t)SCAN_B) lv.push_back((lv.back() + SCAN_B - 1) / SCAN_B);
    size_t tot = 0; for (size_t i = 1; i < lv.size(); i++) tot += lv[i];
    int rc = tmp.reserve((tot + 1) * sizeof(u32)); if (rc) return rc;
    std::vector<u32*> bufs(lv.size()); bufs[0] = out;
    { u32* p = tmp.as<u32>(); for (size_t i = 1; i < lv.size(); i++) { bufs[i] = p; p += lv[i]; } }
    for (size_t l = 0; l < lv.size(); l++) {
        size_t nb = (lv[l] + SCAN_B - 1) / SCAN_B;
        const u32* src = (l == 0) ? in : bufs[l];
        hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(SCAN_T), 0, s, src, bufs[l], (l + 1 < lv.size()) ? bufs[l + 1] : (u32*)nullptr, lv[l]);
    }
    for (size_t l = lv.size() - 1; l-- > 0;) {
        size_t nb = (lv[l] + SCAN_B - 1) / SCAN_B;
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_T), 0, s, bufs[l], bufs[l + 1], lv[l]);
    }
    KH_HIP(hipGetLastError());
    return KH_OK;
}

// ------------------------------------------------------------------------------------ 1 digits
// s: the scalar, Montgomery (mont != 0) or canonical with at most one excess p; digits d_w in (-2^(c-1), 2^(c-1)] with sum_w d_w 2^(c w) = s,
// written to dst[w * stride], w < W (all zero when `skip`)
template <class SF>
__device__ __forceinline__ void emit_digits(Fe<SF> s, int mont, int c, int W, bool skip, int32_t* __restrict__ dst, size_t stride) {
    if (mont) s = from_mont<SF>(s);
    else s = cond_sub_p<SF>(s.v);
    u32 l[8];
#pragma unroll
    for (int t = 0; t < 8; t++) l[t] = s.v[t];
    const u32 half = 1u << (c - 1), mask = (1u << c) - 1u;
    u32 carry = 0;
    for (int w = 0; w < W; w++) {
        u32 v = (l[0] & mask) + carry;
        int32_t d;
        if (v > half) { d = (int32_t)v - (int32_t)(1u << c); carry = 1; } else { d = (int32_t)v; carry = 0; }
#pragma unroll
        for (int t = 0; t < 7; t++) l[t] = (l[t] >> c) | (l[t + 1] << (32 - c));
        l[7] >>= c;
        dst[(size_t)w * stride] = skip ? 0 : d;
    }
}
// ---- GLV split for tables that hold 2^(c w) P and phi(2^(c w) P) (the opening's folded basis, csrc/rebase.hip): k = k1 + k2 lambda (mod r), |k1|, |k2| < 2^127,
// so the window tables need only the doublings of the lower 128 bits -- the chain that builds them is half as long.  The identity holds for ANY rounding of
// c1, c2 (the basis vectors are lattice points: tools/gen_glv_params.py); the rounding only decides the size of k1, k2 (largest seen: 2^126.8).
#include "glv_params.inc"
// c = (k g + 2^383) >> 384, k of 8 limbs, g of 9: five limbs (< 2^130)
__device__ __forceinline__ void glv_mulshift(const u32 k[8], const u32 g[9], u32 c[5]) {
    u32 prod[17];
#pragma unroll
    for (int t = 0; t < 17; t++) prod[t] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const u64 t = (u64)k[i] * g[j] + prod[i + j] + carry;
            prod[i + j] = (u32)t; carry = t >> 32;
        }
        prod[i + 9] = (u32)carry;
    }
    u64 t = (u64)prod[11] + 0x80000000u;                   // + 2^383: round to nearest
    u32 carry = (u32)(t >> 32);
#pragma unroll
    for (int i = 12; i < 17; i++) { t = (u64)prod[i] + carry; c[i - 12] = (u32)t; carry = (u32)(t >> 32); }
}
// acc (8 limbs, modulo 2^256) +/-= x (5 limbs) * y (4 limbs)
__device__ __forceinline__ void glv_muladd(u32 acc[8], const u32 x[5], const u32 y[4], bool add) {
    u32 p[8];
#pragma unroll
    for (int t = 0; t < 8; t++) p[t] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i + j < 8) { const u64 t = (u64)x[i] * y[j] + p[i + j] + carry; p[i + j] = (u32)t; carry = t >> 32; }
        }
        if (i + 4 < 8) p[i + 4] = (u32)carry;
    }
    if (add) { u64 c = 0; _Pragma("unroll") for (int t = 0; t < 8; t++) { const u64 v = (u64)acc[t] + p[t] + c; acc[t] = (u32)v; c = v >> 32; } }
    else { u64 b = 0; _Pragma("unroll") for (int t = 0; t < 8; t++) { const u64 v = (u64)acc[t] - p[t] - b; acc[t] = (u32)v; b = (v >> 32) & 1u; } }
}
// two's complement (8 limbs) -> magnitude (low 4 limbs; the upper four are zero for every scalar: |k_i| < 2^127) and sign
__device__ __forceinline__ bool glv_abs(u32 v[8], u32 mag[4]) {
    const bool neg = (v[7] >> 31) != 0;
    if (neg) { u64 c = 1; _Pragma("unroll") for (int t = 0; t < 8; t++) { const u64 w = (u64)(~v[t]) + c; v[t] = (u32)w; c = w >> 32; } }
#pragma unroll
    for (int t = 0; t < 4; t++) mag[t] = v[t];
    return neg;
}
template <class SF>
__device__ __forceinline__ void glv_split(const u32 k[8], u32 k1[4], bool& n1, u32 k2[4], bool& n2) {
    const GlvK& G = glv_k<SF>();
    u32 g1[9], g2[9], a1[4], b1[4], a2[4], b2[4];
#pragma unroll
    for (int t = 0; t < 9; t++) { g1[t] = G.g1[t]; g2[t] = G.g2[t]; }
#pragma unroll
    for (int t = 0; t < 4; t++) { a1[t] = G.a1[t]; b1[t] = G.b1[t]; a2[t] = G.a2[t]; b2[t] = G.b2[t]; }
    const u32 neg = G.neg;
    const bool a1n = neg & 1u, b1n = (neg >> 1) & 1u, a2n = (neg >> 2) & 1u, b2n = (neg >> 3) & 1u, g1n = (neg >> 4) & 1u, g2n = (neg >> 5) & 1u;
    u32 c1[5], c2[5];
    glv_mulshift(k, g1, c1); glv_mulshift(k, g2, c2);      // |c1|, |c2|; their signs: g1n, g2n (k >= 0)
    u32 t1[8], t2[8];
#pragma unroll
    for (int t = 0; t < 8; t++) { t1[t] = k[t]; t2[t] = 0; }
    // k1 = k - c1 a1 - c2 a2: a product whose sign is negative is ADDED; k2 = -(c1 b1 + c2 b2) likewise
    glv_muladd(t1, c1, a1, g1n != a1n); glv_muladd(t1, c2, a2, g2n != a2n);
    glv_muladd(t2, c1, b1, g1n != b1n); glv_muladd(t2, c2, b2, g2n != b2n);
    n1 = glv_abs(t1, k1); n2 = glv_abs(t2, k2);
}
// W = 2 Wh rows: Wh signed c-bit digits of k1 (rows 0 .. Wh-1, over the tables 2^(c w) P) and Wh of k2 (rows Wh .. 2 Wh - 1, over phi of them)
template <class SF>
__device__ __forceinline__ void emit_digits_glv(Fe<SF> s, int mont, int c, int Wh, bool skip, int32_t* __restrict__ dst, size_t stride) {
    if (mont) s = from_mont<SF>(s);
    else s = cond_sub_p<SF>(s.v);
    u32 k[8], mag[2][4]; bool neg[2];
#pragma unroll
    for (int t = 0; t < 8; t++) k[t] = s.v[t];
    glv_split<SF>(k, mag[0], neg[0], mag[1], neg[1]);
    const u32 half = 1u << (c - 1), mask = (1u << c) - 1u;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        u32 l[5] = {mag[h][0], mag[h][1], mag[h][2], mag[h][3], 0u};
        u32 carry = 0;
        for (int w = 0; w < Wh; w++) {
            const u32 v = (l[0] & mask) + carry;
            int32_t d;
            // the digits that reach the sort lie in (-2^(c-1), 2^(c-1)] as emit_digits' do (a NEGATIVE digit of magnitude 2^(c-1) would pack, at c = 16, into the
            // sort's 0xffff = "no entry"): a magnitude that will be negated takes its own digits from [-2^(c-1), 2^(c-1))
            if (neg[h] ? v >= half : v > half) { d = (int32_t)v - (int32_t)(1u << c); carry = 1; } else { d = (int32_t)v; carry = 0; }
#pragma unroll
            for (int t = 0; t < 4; t++) l[t] = (l[t] >> c) | (l[t + 1] << (32 - c));
            dst[(size_t)(h * Wh + w) * stride] = skip ? 0 : (neg[h] ? -d : d);
        }
    }
}
template <class SF>
__global__ void k_digits_glv(const u64* __restrict__ scalars, const uint8_t* __restrict__ inf, size_t inf_off,
                             size_t inf_batch, size_t n, int mont, int c, int W, int32_t* __restrict__ digits) {
    KH_HIGH_PRIO();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= n) return;
    const bool skip = inf && inf[inf_off + (size_t)j * inf_batch + i];
    emit_digits_glv<SF>(Fe<SF>::load(scalars + ((size_t)j * n + i) * 4), mont, c, W / 2, skip, digits + (size_t)j * W * n + i, n);
}
// test hook: the split of n canonical scalars (kh_debug_glv_split)
template <class SF>
__global__ void k_glv_split_test(const u64* __restrict__ scalars, size_t n, u32* __restrict__ out /* n x (4 + 4 + 2) words */) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe<SF> s = Fe<SF>::load(scalars + 4 * i);
    u32 k[8], m1[4], m2[4]; bool n1, n2;
#pragma unroll
    for (int t = 0; t < 8; t++) k[t] = s.v[t];
    glv_split<SF>(k, m1, n1, m2, n2);
#pragma unroll
    for (int t = 0; t < 4; t++) { out[10 * i + t] = m1[t]; out[10 * i + 4 + t] = m2[t]; }
    out[10 * i + 8] = n1 ? 1u : 0u; out[10 * i + 9] = n2 ? 1u : 0u;
}
int msm_debug_glv_split(hipStream_t s, int field, const uint64_t* scalars_dev, size_t n, uint32_t* out_dev) {
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_glv_split_test<FpParams>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scalars_dev, n, out_dev);
    else hipLaunchKernelGGL((k_glv_split_test<FqParams>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scalars_dev, n, out_dev);
    KH_HIP(hipGetLastError());
    return KH_OK;
}
const uint32_t* msm_glv_lambda(int scalar_field) { return scalar_field == KH_FIELD_FP ? GLV_HOST_FP.lambda : GLV_HOST_FQ.lambda; }
template <class SF>
__global__ void k_digits(const u64* __restrict__ scalars, const uint8_t* __restrict__ inf, size_t inf_off,
                         size_t inf_batch, size_t n, int mont, int c, int W, int32_t* __restrict__ digits, size_t i0, size_t i1) {
    KH_HIGH_PRIO();
    size_t i = i0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // scalars [i0, i1) of the n (a chunk of an upload still in flight, or all of them)
    int j = blockIdx.y;
    if (i >= i1) return;
    const bool skip = inf && inf[inf_off + (size_t)j * inf_batch + i];
    emit_digits<SF>(Fe<SF>::load(scalars + ((size_t)j * n + i) * 4), mont, c, W, skip, digits + (size_t)j * W * n + i, n);     // (canonical input may carry one excess p)
}

// ------------------------------------------------------------------------------------ 2 histogram
struct SortGeom {
    size_t n;       // scalars per MSM
    u32 nb;         // buckets per group = 2^(c-1)
    int S;          // slices per (window, msm)
    int W;          // windows
    int precomp;    // 1: all windows of an MSM share one bucket group
    size_t pt_stride;   // precomp: points per window table
    size_t pt_offset;   // first basis point used
    size_t pt_batch;    // index step between the bases of consecutive MSMs of a batch (0: shared basis)
};
__device__ __forceinline__ void geom_ids(const SortGeom& g, int s, int w, int j, size_t& q, int& Sq, int& sigma) {
    if (g.precomp) { q = (size_t)j; Sq = g.W * g.S; sigma = w * g.S + s; }
    else { q = (size_t)j * g.W + w; Sq = g.S; sigma = s; }
}
__global__ void k_hist(const int32_t* __restrict__ digits, SortGeom g, u32* __restrict__ H) {
    KH_HIGH_PRIO();
    extern __shared__ u32 h[];
    int s = blockIdx.x, w = blockIdx.y, j = blockIdx.z;
    for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) h[b] = 0;
    __syncthreads();
    size_t lo = g.n * (size_t)s / g.S, hi = g.n * (size_t)(s + 1) / g.S;
    const int32_t* d = digits + ((size_t)j * g.W + w) * g.n;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        int32_t v = d[i];
        if (v) atomicAdd(&h[(v < 0 ? -v : v) - 1], 1u);
    }
    __syncthreads();
    size_t q; int Sq, sigma; geom_ids(g, s, w, j, q, Sq, sigma);
    u32* out = H + (q * Sq + sigma) * g.nb;
    for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) out[b] = h[b];
}
// per key: turn the per-slice counts into exclusive within-key prefixes, emit the key total
__global__ void k_key_totals(u32* __restrict__ H, u32 nb, int Sq, size_t nkeys, u32* __restrict__ cnt) {
    KH_HIGH_PRIO();
    size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) { if (key == nkeys) cnt[key] = 0; return; }
    size_t q = key / nb; u32 b = (u32)(key % nb);
    u32* p = H + q * Sq * nb + b;
    u32 run = 0;
    for (int s = 0; s < Sq; s++) { u32 v = p[(size_t)s * nb]; p[(size_t)s * nb] = run; run += v; }
    cnt[key] = run;
}
// ------------------------------------------------------------------------------------ 4 scatter
__global__ void k_scatter(const int32_t* __restrict__ digits, SortGeom g, const u32* __restrict__ H,
                          const u32* __restrict__ off, u32* __restrict__ entries) {
    KH_HIGH_PRIO();
    extern __shared__ u32 pos[];
    int s = blockIdx.x, w = blockIdx.y, j = blockIdx.z;
    size_t q; int Sq, sigma; geom_ids(g, s, w, j, q, Sq, sigma);
    const u32* hin = H + (q * Sq + sigma) * g.nb;
    const u32* o = off + q * g.nb;
    for (u32 b = threadIdx.x; b < g.nb; b += blockDim.x) pos[b] = o[b] + hin[b];
    __syncthreads();
    size_t lo = g.n * (size_t)s / g.S, hi = g.n * (size_t)(s + 1) / g.S;
    const int32_t* d = digits + ((size_t)j * g.W + w) * g.n;
    u32 pbase = (u32)(g.pt_offset + (g.precomp ? (size_t)w * g.pt_stride : 0) + (size_t)j * g.pt_batch);
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        int32_t v = d[i];
        if (v) {
            u32 b = (u32)(v < 0 ? -v : v) - 1u;
            u32 slot = atomicAdd(&pos[b], 1u);
            entries[slot] = (pbase + (u32)i) | (v < 0 ? 0x80000000u : 0u);
        }
    }
}
// ------------------------------------------------------------------------------------ partitioned sort (large jobs over the tables)
// k_scatter writes each 4-byte entry to a random place in a 67 MB array: every store is its own partial cache line (PMC: 538 MB of
// write traffic for 67 MB of payload at 2^20 points, and the slowest kernel of the sort).  For the 2^15 buckets of the window
// tables the sort runs in two passes instead, each with few enough open output streams per block that a line is completed in
// the L2 before it is evicted:
//   pass A  by the top 8 bits of the bucket (256 partitions): k_part_hist counts per (partition, block), one scan gives every
//           block its output run inside every partition, k_part_scatter writes 32-bit records
//           (7 low bucket bits | sign | window | point) through 256 LDS cursors -- 256 sequential streams per block;
//   pass B  one block per partition (~64 K records, read twice from the L2): histogram of its 128 buckets, off[], then the
//           final entries through 128 LDS cursors into a 256 KB window.
// No per-slice histogram matrix, no key-total pass, no global scan over the keys.  Needs n <= 2^20 and W <= 16 for the record.
static constexpr int PART_T = 1024, PART_P = 256, PART_LOW = 7;
// LDS counter increment with wave aggregation for the skewed case: when every active lane of the wave targets the SAME counter (the
// reference's benchmark witness puts n - 10 of n scalars into one bucket: 65,526 same-address LDS atomics serialise, 228 us in
// k_part_sort for 15 such columns) ONE lane adds the lane count and the others take consecutive positions; a mixed wave pays one
// ballot + one shuffle more than the plain per-lane atomics it then issues.  Must be called by all non-exited lanes of the wave together.
__device__ __forceinline__ u32 lds_inc_agg(u32* ctr, u32 key, bool active) {
    const u64 act = __ballot(active);
    if (!act) return 0;
    const int leader = __ffsll((long long)act) - 1;
    const u32 k0 = __shfl(key, leader, 64);
    if (__ballot(active && key == k0) == act) {
        const int lane = (int)(threadIdx.x & 63u);
        u32 b = 0;
        if (lane == leader) b = atomicAdd(&ctr[k0], (u32)__popcll(act));
        b = __shfl(b, leader, 64);
        return b + (u32)__popcll(act & ((1ull << lane) - 1ull));
    }
    return active ? atomicAdd(&ctr[key], 1u) : 0u;
}
__global__ void __launch_bounds__(PART_T)
k_part_hist(const int32_t* __restrict__ digits, u32 tot_e, u32 nblk, u32 shift, u32* __restrict__ PH) {
    KH_HIGH_PRIO();
    __shared__ u32 cnt[PART_P];
    const u32 blk = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    if (tid < PART_P) cnt[tid] = 0;
    __syncthreads();
    const u32 e_lo = (u32)((u64)tot_e * blk / nblk), e_hi = (u32)((u64)tot_e * (blk + 1) / nblk);
    const int32_t* d = digits + (size_t)j * tot_e;
    for (u32 e0 = e_lo + tid; e0 < e_hi; e0 += 8 * PART_T) {
        int32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 e = e0 + u * PART_T; v[u] = d[e < e_hi ? e : e_hi - 1]; if (e >= e_hi) v[u] = 0; }
#pragma unroll
        for (int u = 0; u < 8; u++) (void)lds_inc_agg(cnt, (((u32)(v[u] < 0 ? -v[u] : v[u]) - 1u) >> shift) & (PART_P - 1u), v[u] != 0);
    }
    __syncthreads();
    if (tid < PART_P) PH[((size_t)j * PART_P + tid) * nblk + blk] = cnt[tid];
    if (blk == 0 && j == 0 && tid == 0) PH[(size_t)gridDim.y * PART_P * nblk] = 0;      // the scan runs one element further: the total
}
__global__ void __launch_bounds__(PART_T)
k_part_scatter(const int32_t* __restrict__ digits, u32 tot_e, u32 n, u32 nblk, const u32* __restrict__ PO, u32* __restrict__ mid) {
    KH_HIGH_PRIO();
    __shared__ u32 cur[PART_P];
    const u32 blk = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    if (tid < PART_P) cur[tid] = PO[((size_t)j * PART_P + tid) * nblk + blk];
    __syncthreads();
    const u32 e_lo = (u32)((u64)tot_e * blk / nblk), e_hi = (u32)((u64)tot_e * (blk + 1) / nblk);
    const int32_t* d = digits + (size_t)j * tot_e;
    u32 w = (e_lo + tid) / n, i = (e_lo + tid) - w * n;
    for (u32 e0 = e_lo + tid; e0 < e_hi; e0 += 8 * PART_T) {
        int32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 e = e0 + u * PART_T; v[u] = d[e < e_hi ? e : e_hi - 1]; if (e >= e_hi) v[u] = 0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const u32 b = (u32)(v[u] < 0 ? -v[u] : v[u]) - 1u;
            const u32 pos = lds_inc_agg(cur, b >> PART_LOW, v[u] != 0);
            if (v[u]) mid[pos] = ((b & ((1u << PART_LOW) - 1u)) << 25) | (v[u] < 0 ? 1u << 24 : 0u) | (w << 20) | i;
            i += PART_T;
            while (i >= n) { i -= n; w++; }
        }
    }
}
__global__ void __launch_bounds__(PART_T)
k_part_sort(const u32* __restrict__ mid, const u32* __restrict__ PO, u32 nblk, u32 nb, size_t pt_stride, size_t pt_offset, size_t pt_batch,
            u32 last_group, u32* __restrict__ off, u32* __restrict__ entries) {
    KH_HIGH_PRIO();
    __shared__ u32 hist[1u << PART_LOW], cur[1u << PART_LOW];
    const u32 pidx = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const size_t q = (size_t)j * PART_P + pidx;
    const u32 base = PO[q * nblk], end = PO[(q + 1) * nblk];          // (the scan has one element more than the matrix: the total)
    if (tid < (1u << PART_LOW)) hist[tid] = 0;
    __syncthreads();
    for (u32 x0 = base + tid; x0 < end; x0 += 8 * PART_T) {
        u32 m[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 x = x0 + u * PART_T; m[u] = mid[x < end ? x : end - 1]; }
#pragma unroll
        for (int u = 0; u < 8; u++) (void)lds_inc_agg(hist, m[u] >> 25, x0 + u * PART_T < end);
    }
    __syncthreads();
    if (tid < 64) {                                      // exclusive scan of the 128 counts: two per lane of one wave
        const u32 a = hist[2 * tid], b = hist[2 * tid + 1];
        u32 inc = a + b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d, 64); if (tid >= (u32)d) inc += o; }
        const u32 ex = base + inc - (a + b);
        u32* o = off + (size_t)j * nb + (size_t)pidx * (1u << PART_LOW);
        o[2 * tid] = ex; o[2 * tid + 1] = ex + a;
        cur[2 * tid] = ex; cur[2 * tid + 1] = ex + a;
        if (tid == 63 && pidx == PART_P - 1 && j == last_group) off[(size_t)(j + 1) * nb] = end;       // off[nkeys] = number of entries
    }
    __syncthreads();
    const u32 pb0 = (u32)(pt_offset + (size_t)j * pt_batch);
    for (u32 x0 = base + tid; x0 < end; x0 += 8 * PART_T) {
        u32 m[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 x = x0 + u * PART_T; m[u] = mid[x < end ? x : end - 1]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool live = x0 + u * PART_T < end;
            const u32 pos = lds_inc_agg(cur, m[u] >> 25, live);
            if (live) entries[pos] = (pb0 + (u32)(((m[u] >> 20) & 15u) * pt_stride) + (m[u] & 0xfffffu)) | ((m[u] & (1u << 24)) << 7);
        }
    }
}
// ------------------------------------------------------------------------------------ tasks
static constexpr u32 MAX_K = 256;           // upper bound of the task length (length bins of the task ordering)
// K (entries per task) is chosen HERE from the actual number of entries off[nkeys]: zero digits
// produce no entry, and the reference's own benchmark witness is almost all ones (one non-zero
// digit per scalar) -- sizing K from the n*W upper bound left the chip 5 % occupied on it.
// For SMALL jobs (a few blocks per CU) the task length also decides how the blocks quantise onto the chip: k_accumulate29 is issue-bound, a CU that
// gets three blocks where its neighbours get two makes the whole launch take three task-lengths.  Measured on the L / R pair of an opening round
// (tools/kmin_sweep.sh; 2^16: K = 8 / 10 / 11 -> 5.68 / 5.88 / 5.47 ms of rounds, 2^14: 4.25 / 3.73 / 3.74) and reproduced by the model
//     cost(K) = ceil(blocks(K) / CUs) * min(K, m + 2.3 sqrt(m)) + 2 * tasks-per-bucket(K),   bucket sizes ~ Poisson(m), m = entries / buckets
// (the second term: every further task of a bucket is one more partial for the bucket sums).  The host tabulates argmin_K over m in quarter octaves
// for the launch's bucket count (KTab, 0 = no opinion: a big job, where only the total matters); the device looks m up from the ACTUAL entry count.
struct KTab { uint8_t k[48]; };
__device__ __forceinline__ u32 pick_K(u32 total, u32 room, u32 kmin, const KTab& tab, u32 nkeys) {
    u32 K = (total + room - 1) / room;
    u32 lo = kmin;
    if (total && nkeys) {
        const float m = (float)total / (float)nkeys;
        int idx = (int)(4.0f * __log2f(m) + 8.5f);
        idx = idx < 0 ? 0 : (idx > 47 ? 47 : idx);
        if (tab.k[idx]) lo = tab.k[idx];
    }
    return K < lo ? lo : (K > MAX_K ? MAX_K : K);
}
// the length histogram of the task ordering and the hot-bucket list's two counters, emptied for the launch sequence that follows
__global__ void k_task_reset(u32* __restrict__ len_hist, u32* __restrict__ big) {
    KH_HIGH_PRIO();
    if (threadIdx.x <= MAX_K) len_hist[threadIdx.x] = 0;
    if (threadIdx.x < 2) big[threadIdx.x] = 0;
}
__global__ void k_ntask(const u32* __restrict__ off, size_t nkeys, u32 room, u32 kmin, KTab ktab, u32* __restrict__ nt, u32* __restrict__ len_hist, u32* __restrict__ handed) {
    KH_HIGH_PRIO();
    __shared__ u32 h[MAX_K + 1];
    if (handed && blockIdx.x == 0 && threadIdx.x == 0) handed[0] = 0;          // hand-over list of the accumulation that follows
    for (u32 i = threadIdx.x; i <= MAX_K; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const u32 K = pick_K(off[nkeys], room, kmin, ktab, (u32)nkeys);
    size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (key <= nkeys) {
        u32 n_t = 0;
        if (key < nkeys) {
            u32 c = off[key + 1] - off[key];
            n_t = (c + K - 1) / K;
            u32 len = n_t ? (c + n_t - 1) / n_t : 0;
            atomicAdd(&h[len], 1u);
        }
        nt[key] = n_t;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i <= MAX_K; i += blockDim.x) if (h[i]) atomicAdd(&len_hist[i], h[i]);
}
// Task ordering: keys are ranked by the length of their tasks, longest first, so that the 64 lanes of
// a wave run chains of (nearly) the same length (bucket sizes are Poisson-distributed: +-20 % at 32
// entries per bucket) and the short tasks fill the end of the launch.
// len_hist -> cursor[L] = first rank of length L (descending order)
__global__ void __launch_bounds__(320) k_len_starts(const u32* __restrict__ len_hist, u32* __restrict__ cursor) {
    KH_HIGH_PRIO();
    // thread L: the number of keys with a longer task (one thread walking the 257 bins took 11 us of every batched MSM)
    __shared__ u32 h[MAX_K + 1];
    if (threadIdx.x <= MAX_K) h[threadIdx.x] = len_hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x <= MAX_K) {
        u32 run = 0;
        for (u32 L = threadIdx.x + 1; L <= MAX_K; L++) run += h[L];
        cursor[threadIdx.x] = run;
    }
}
__global__ void __launch_bounds__(1024)
k_len_rank(const u32* __restrict__ off, const u32* __restrict__ nt, size_t nkeys,
           u32* __restrict__ cursor, u32* __restrict__ order, u32* __restrict__ rnt) {
    KH_HIGH_PRIO();
    // two levels: rank inside the block with LDS atomics, then ONE global cursor increment per
    // (block, length) -- per-wave increments serialise on the few hot lengths (measured 0.6 ms)
    __shared__ u32 cnt[MAX_K + 1], base[MAX_K + 1];
    for (u32 i = threadIdx.x; i <= MAX_K; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool active = key < nkeys;
    u32 n_t = 0, len = 0, local = 0;
    if (active) {
        n_t = nt[key];
        u32 c = off[key + 1] - off[key];
        len = n_t ? (c + n_t - 1) / n_t : 0;
        local = atomicAdd(&cnt[len], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i <= MAX_K; i += blockDim.x) if (cnt[i]) base[i] = atomicAdd(&cursor[i], cnt[i]);
    __syncthreads();
    if (active) { u32 rank = base[len] + local; order[rank] = (u32)key; rnt[rank] = n_t; }
    if (key == nkeys) rnt[nkeys] = 0;
}
// ------------------------------------------------------------------------------------ fused sort (small jobs over the tables)
// One to four MSMs of <= ~2^17 scalars over the window tables -- the L / R pair of an opening round, a lone commitment -- are
// launch-bound in the sort: histogram, key totals, two three-level scans, task counts, scatter and two memsets were 13 dependent
// launches of a few microseconds of work each (~0.11 ms of the 0.49 ms round).  k_sort_fused runs all of it in ONE launch of
// <= 32 co-resident blocks separated by three grid barriers (an agent-scope counter in global memory; the blocks of four jobs in
// flight fill at most half of the CUs, so every block is resident and the spin cannot deadlock).  Block (j, r) owns the r-th
// contiguous chunk of group j's digit matrix [w][i]; its LDS histogram is one "slice" of the deterministic counting sort, the
// keys are scanned with one (thread-contiguous keys, block scan, block totals) pass per barrier, and the same block scatters its
// chunk.  Outputs are exactly those of the multi-launch path (off / toff / entries, empty hand-over and big-bucket lists).
struct FusedGeom {
    u32 n, nb, W, k, bpg, split, sub, kpt, nkeys, room, kmin;     // sub = nb / split buckets per block
    KTab ktab;
    size_t pt_stride, pt_offset, pt_batch;
};
static constexpr int FUSED_T = 1024, FUSED_KPT = 2, FUSED_G = 16, FUSED_B = 64;      // chunks per job <= FUSED_G, blocks <= FUSED_B
static constexpr int FUSED_V = 17, FUSED_C = 16;        // int4's of digits / cursors per thread (registers)
// Bounded: the barrier assumes every block of the launch is resident.  Inside one process that holds (64 blocks of 64 KB LDS, two per
// CU, even with all four pipeline slots in flight); with several PROCESSES on one GPU it need not, and resident blocks would spin on
// peers that cannot be scheduled.  After `limit` ticks of the 100 MHz wall clock (or when another block has given up) the barrier
// returns false, the whole launch drains, and the host re-runs the job with the multi-launch sort (msm_finish).
__device__ __forceinline__ bool grid_barrier(u32* ctr, u32 target, u32* abort_dev, volatile uint32_t* abort_host, unsigned long long limit) {
    __shared__ int ok_;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        const unsigned long long t0 = wall_clock64();
        u32 spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((spins++ & 127u) == 0 && (__hip_atomic_load(abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > limit)) {
                __hip_atomic_store(abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *abort_host = 1u;
                ok = 0;
                break;
            }
        }
        ok_ = ok;
    }
    __syncthreads();
    return ok_ != 0;
    // no acquire fence: an agent-scope invalidate of the L2 cost ~7 us per barrier (every wave's first load afterwards waited for
    // it).  Everything another block wrote is read with coherent (agent-scope, sc1) loads instead: coh_load below.
}
__device__ __forceinline__ u32 coh_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A k_sort_fused launch that gave up at a barrier leaves off / toff / entries / the big-bucket list partly written -- on the first job of a
// process that is uninitialised hipMalloc memory, and a garbage 31-bit entry index would read far out of bounds before the host ever sees
// *host_abort.  Every kernel queued behind a fused sort that INDEXES through those arrays therefore gets the sort's abort word and leaves at
// once when it is set (the marginal kernels read fixed ranges of the bucket array: garbage in, garbage out, and msm_finish redoes the job).
__device__ __forceinline__ bool sort_gave_up(const u32* abort_dev) { return abort_dev && coh_load(abort_dev) != 0u; }
// exclusive prefix of v over the block's 1024 threads; *total = the block sum (sh: >= 16 words)
__device__ __forceinline__ u32 block_scan_1024(u32 v, u32* sh, u32* total) {
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u32 o = __shfl_up(inc, d, 64); if (lane >= (u32)d) inc += o; }
    __syncthreads();                                     // sh may still be read from the previous call
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (u32 w2 = 0; w2 < FUSED_T / 64; w2++) { const u32 t = sh[w2]; if (w2 < wave) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}
// sum of bs[0 .. count) and of bs[0 .. upto) with ONE memory round trip (lane l loads bs[l]; count <= 64)
__device__ __forceinline__ void lane_sums(const u32* bs, u32 count, u32 upto, u32* below, u32* total) {
    const u32 lane = threadIdx.x & 63u;
    u32 v = coh_load(bs + (lane < count ? lane : 0u));
    if (lane >= count) v = 0u;
    u32 lo = lane < upto ? v : 0u, all = v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo += __shfl_xor(lo, d, 64); all += __shfl_xor(all, d, 64); }
    *below = lo; *total = all;
}
__global__ void __launch_bounds__(FUSED_T)
k_sort_fused(const int32_t* __restrict__ digits, FusedGeom g, u32* __restrict__ H, u32* __restrict__ off, u32* __restrict__ toff,
             u32* __restrict__ entries, u32* __restrict__ handed, u32* __restrict__ big, u32* __restrict__ sync, u32* __restrict__ bsums,
             volatile uint32_t* abort_host, unsigned long long spin_limit) {
    KH_HIGH_PRIO();
    extern __shared__ u32 lds[];                         // `sub` words: histogram, later the scatter cursors
    __shared__ u32 sh[FUSED_T / 64 + 1];
    const u32 G = gridDim.x, blk = blockIdx.x, tid = threadIdx.x;
    // block = (MSM j of the batch, chunk r of its digit matrix [w][i], bucket sub-range h of that chunk)
    const u32 h = blk % g.split, r = (blk / g.split) % g.bpg, j = blk / (g.split * g.bpg);
    const u32 b_lo = h * g.sub;
    if (blk == 0 && tid == 0) { handed[0] = 0; big[0] = 0; big[1] = 0; }
#define KH_TS(i) do { if (blk == 0 && tid == 0) ((unsigned long long*)(bsums + 2 * FUSED_B))[i] = wall_clock64(); } while (0)     // KH_FUSED_DEBUG prints them
    KH_TS(0);
    // 1 the chunk's digits into registers (every load of this kernel misses the L2 -- the data comes from other XCDs -- so all of
    //   a phase's loads are issued before anything waits: the phases are memory round trips, not bandwidth), then the histogram
    const u32 tot4 = (g.W * g.n) / 4;                                   // chunks are whole int4's of the [w][i] matrix
    const u32 q_lo = (u32)((u64)tot4 * r / g.bpg), q_hi = (u32)((u64)tot4 * (r + 1) / g.bpg);
    const int4* d4 = (const int4*)(digits + (size_t)j * g.W * g.n);
    // digits are kept as 16-bit codes, two per register: sign << 15 | (|d| - 1), 0xffff for d = 0 (|d| - 1 = 32767 is never negative)
    u32 dv[FUSED_V][2];
#pragma unroll
    for (int t0 = 0; t0 < FUSED_V; t0 += 6) {                           // six 16-byte loads in flight per thread
        int4 x[6];
#pragma unroll
        for (int u = 0; u < 6; u++) if (t0 + u < FUSED_V) {            // unconditional loads (clamped index): a predicated load is waited for on the spot
            const u32 q = q_lo + tid + (t0 + u) * FUSED_T;
            x[u] = d4[q < q_hi ? q : q_hi - 1];
        }
#pragma unroll
        for (int u = 0; u < 6; u++) if (t0 + u < FUSED_V) {
            const bool in = q_lo + tid + (t0 + u) * FUSED_T < q_hi;
            const int32_t v[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
            u32 c[4];
#pragma unroll
            for (int l = 0; l < 4; l++) c[l] = (!in || v[l] == 0) ? 0xffffu : ((v[l] < 0 ? 0x8000u : 0u) | ((u32)(v[l] < 0 ? -v[l] : v[l]) - 1u));
            dv[t0 + u][0] = c[0] | (c[1] << 16); dv[t0 + u][1] = c[2] | (c[3] << 16);
        }
    }
    for (u32 b = tid; b < g.sub; b += FUSED_T) lds[b] = 0;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < FUSED_V; t++) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const u32 c = (dv[t][l >> 1] >> (16 * (l & 1))) & 0xffffu, b = (c & 0x7fffu) - b_lo;
            if (c != 0xffffu && b < g.sub) atomicAdd(&lds[b], 1u);
        }
    }
    __syncthreads();
    u32* hout = H + ((size_t)j * g.bpg + r) * g.nb + b_lo;
    for (u32 b = tid; b < g.sub; b += FUSED_T) hout[b] = lds[b];
    KH_TS(1);
    u32* const abort_dev = bsums + 2 * FUSED_B + 32;
    if (!grid_barrier(sync, G, abort_dev, abort_host, spin_limit)) return;
    KH_TS(2);
    // 2 per key: within-key prefixes over the chunks, the key's count; block-local scan of the counts
    const u32 key0 = (blk * FUSED_T + tid) * g.kpt;
    u32 cnt[FUSED_KPT], mine = 0;
    {
        u32 hv[FUSED_KPT][FUSED_G];                          // 32-bit indices from the uniform base: one offset register per access
#pragma unroll
        for (int i = 0; i < FUSED_KPT; i++) {
            const u32 key = key0 + i;
            const bool live = (u32)i < g.kpt && key < g.nkeys;
            const u32 idx = (key / g.nb) * g.bpg * g.nb + key % g.nb;
#pragma unroll
            for (int c = 0; c < FUSED_G; c++) hv[i][c] = coh_load(H + (live ? idx : 0u) + ((u32)c < g.bpg ? (u32)c : 0u) * g.nb);     // unconditional, masked below
#pragma unroll
            for (int c = 0; c < FUSED_G; c++) if (!(live && (u32)c < g.bpg)) hv[i][c] = 0u;
        }
#pragma unroll
        for (int i = 0; i < FUSED_KPT; i++) {
            const u32 key = key0 + i;
            const bool live = (u32)i < g.kpt && key < g.nkeys;
            const u32 idx = (key / g.nb) * g.bpg * g.nb + key % g.nb;
            u32 run = 0;
#pragma unroll
            for (int c = 0; c < FUSED_G; c++) { if (live && (u32)c < g.bpg) H[idx + (u32)c * g.nb] = run; run += hv[i][c]; }
            cnt[i] = run; mine += run;
        }
    }
    u32 btot;
    const u32 ex1 = block_scan_1024(mine, sh, &btot);
    if (tid == 0) bsums[blk] = btot;
    KH_TS(3);
    if (!grid_barrier(sync, 2 * G, abort_dev, abort_host, spin_limit)) return;
    KH_TS(4);
    // 3 off[]; the task length K from the true entry count; task counts and their block-local scan
    u32 base, total;
    lane_sums(bsums, G, blk, &base, &total);
    KH_TS(9);
    const u32 K = pick_K(total, g.room, g.kmin, g.ktab, g.nkeys);
    u32 run = base + ex1, ntm = 0, nt[FUSED_KPT];
#pragma unroll
    for (int i = 0; i < FUSED_KPT; i++) {
        const u32 key = key0 + i;
        nt[i] = (cnt[i] + K - 1) / K;
        if ((u32)i < g.kpt && key <= g.nkeys) off[key] = run;
        run += cnt[i]; ntm += nt[i];
    }
    KH_TS(10);
    const u32 ex2 = block_scan_1024(ntm, sh, &btot);
    KH_TS(11);
    if (tid == 0) bsums[FUSED_B + blk] = btot;
    KH_TS(5);
    if (!grid_barrier(sync, 3 * G, abort_dev, abort_host, spin_limit)) return;
    KH_TS(6);
    // 4 scatter cursors (key offset + the chunk's within-key prefix); toff[]; scatter of the digits held in registers
    const u32* o = off + (size_t)j * g.nb + b_lo;
    {
        u32 a[FUSED_C], hh[FUSED_C];
#pragma unroll
        for (int u = 0; u < FUSED_C; u++) { const u32 b = tid + u * FUSED_T, bc = b < g.sub ? b : 0u; a[u] = coh_load(o + bc); hh[u] = coh_load(hout + bc); }
        u32 total2;
        lane_sums(bsums + FUSED_B, G, blk, &base, &total2);
#pragma unroll
        for (int u = 0; u < FUSED_C; u++) { const u32 b = tid + u * FUSED_T; if (b < g.sub) lds[b] = a[u] + hh[u]; }
    }
    run = base + ex2;
#pragma unroll
    for (int i = 0; i < FUSED_KPT; i++) {
        const u32 key = key0 + i;
        if ((u32)i < g.kpt && key <= g.nkeys) toff[key] = run;
        run += nt[i];
    }
    __syncthreads();
    KH_TS(7);
    const u32 pb0 = (u32)(g.pt_offset + (size_t)j * g.pt_batch);
    const u32 e_first = 4 * (q_lo + tid);
    u32 w = e_first / g.n, i = e_first - w * g.n;                       // (window, point) of the int4's first digit, advanced incrementally
#pragma unroll
    for (int t = 0; t < FUSED_V; t++) asm volatile("" : "+v"(dv[t][0]), "+v"(dv[t][1]));       // (no values of phase 1 kept alive: they spilled)
#pragma unroll
    for (int t = 0; t < FUSED_V; t++) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const u32 c = (dv[t][l >> 1] >> (16 * (l & 1))) & 0xffffu, b = (c & 0x7fffu) - b_lo;
            if (c != 0xffffu && b < g.sub) {
                u32 ii = i + l, ww = w;
                if (ii >= g.n) { ii -= g.n; ww++; }
                const u32 slot = atomicAdd(&lds[b], 1u);
                entries[slot] = (pb0 + (u32)(ww * g.pt_stride) + ii) | ((c & 0x8000u) << 16);
            }
        }
        i += 4 * FUSED_T;
        while (i >= g.n) { i -= g.n; w++; }
    }
    // the last block out re-arms the barrier counter for the next launch
    __syncthreads();
    KH_TS(8);
    if (tid == 0) {
        const u32 out = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (out == G - 1) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// ------------------------------------------------------------------------------------ partitioned sort, wide windows (c = 20: 2^19 buckets)
// The same two passes for 2^(8 + low) buckets, low <= PART2_MAXLOW: pass B then sorts a partition of 2^low buckets through 2^low LDS cursors.
// A partition is the set of buckets with the same LOW 8 bits here: the top window of a 255-bit scalar holds only 15 bits, so its 2^20 entries fall
// into the 2^14 lowest buckets (+64 entries each, on top of ~24) -- eight contiguous partitions would get 3.5 x the records of the others (pass B took
// 300 us instead of 90) and a length ranking per partition would differ wildly between partitions; with the low bits every partition holds its share
// of the heavy buckets.  Everything downstream of the sort therefore works on PERMUTED keys  key' = (bucket & 255) << low | bucket >> 8  (per MSM);
// only the bucket records the reduction reads are stored by true bucket number (wide_true_bucket).  The
// 32-bit record of pass A cannot carry 11 bucket bits + sign + window + 20 point bits any more; it carries the entry's position INSIDE ITS PASS-A
// BLOCK instead (16 bits: a block reads a contiguous run of <= 65,536 digits of the [w][i] matrix): pass B knows from the scanned block matrix PO
// which block wrote the record it is looking at (the partition's run is the concatenation of the blocks' runs, in block order) and rebuilds
// (window, point) from block start + local position.
static constexpr u32 PART2_MAXLOW = 11, PART2_MAXPASS = 8;  // (array bound; the launch's max_pass decides: 2 by default)
__global__ void __launch_bounds__(PART_T)
k_part2_scatter(const int32_t* __restrict__ digits, u32 tot_e, u32 nblk, u32 low, const u32* __restrict__ PO, u32* __restrict__ mid, u32* __restrict__ xlist, u32* __restrict__ big) {
    KH_HIGH_PRIO();
    __shared__ u32 cur[PART_P];
    const u32 blk = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    if (blk == 0 && j == 0 && tid == 0) { xlist[0] = 0; xlist[1] = 0; big[0] = 0; }      // the split-bucket and hot-bucket lists pass B appends to
    if (tid < PART_P) cur[tid] = PO[((size_t)j * PART_P + tid) * nblk + blk];
    __syncthreads();
    const u32 e_lo = (u32)((u64)tot_e * blk / nblk), e_hi = (u32)((u64)tot_e * (blk + 1) / nblk);
    const int32_t* d = digits + (size_t)j * tot_e;
    for (u32 e0 = e_lo + tid; e0 < e_hi; e0 += 8 * PART_T) {
        int32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 e = e0 + u * PART_T; v[u] = d[e < e_hi ? e : e_hi - 1]; if (e >= e_hi) v[u] = 0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const u32 b = (u32)(v[u] < 0 ? -v[u] : v[u]) - 1u;
            const u32 pos = lds_inc_agg(cur, b & (PART_P - 1u), v[u] != 0);
            if (v[u]) mid[pos] = ((b >> 8) << 17) | (v[u] < 0 ? 1u << 16 : 0u) | (e0 + u * PART_T - e_lo);
        }
    }
}
// Pass B also plans the accumulation: with ~26 entries per bucket a task IS a bucket (buckets above K entries are split as on the narrow path), so the
// task counts, their prefix (toff) and the length ranking are partition-local here (the narrow path ranks all keys globally: two more kernels and
// two more device-wide scans, 64 us at 2^19 keys).  toff is written relative to the partition (k_wide_fixup adds the partitions' task bases);
// the order is INTERLEAVED over the partitions -- order[r * nparts + q] = the bucket of rank r (longest first) in partition q -- so that the
// accumulation launch runs longest-first as a whole (partitions are statistically alike: see the choice of the partition bits above), `og`
// consecutive ranks of a partition side by side (measured: og = 1 / 4 / 16 / 64 / 2048 -> 923 / 843 / 795 / 819 / 1710 us of accumulation).  The extra
// chunks of split buckets go to xlist as (key, chunk) pairs ([0] = their number), the split buckets themselves to slist ([1] = their number);
// empty buckets get their identity record here.
struct WideTasks { u32 room, kmin, nkeys; KTab ktab; };
__global__ void __launch_bounds__(PART_T)
k_part2_sort(const u32* __restrict__ mid, const u32* __restrict__ PO, u32 nblk, u32 low, u32 nb, u32 tot_e, u32 n, size_t pt_stride, size_t pt_offset,
             size_t pt_batch, u32 last_group, size_t total_idx, WideTasks ta, u32* __restrict__ off, u32* __restrict__ entries,
             u32* __restrict__ toff, u32* __restrict__ order, u32* __restrict__ ptot, u32* __restrict__ xlist, u32* __restrict__ slist,
             uint8_t* __restrict__ buckets29, u32 og, u32 stage_cap, u32 max_pass, u32* __restrict__ big, u32 bigcap, u32 hot_nt) {
    KH_HIGH_PRIO();
    __shared__ u32 cur[1u << PART2_MAXLOW], pstart[1025], bi0[1024], bw0[1024], lh[MAX_K + 2], sh[PART_T / 64 + 1];
    __shared__ u32 xl_n, pfirst[PART2_MAXPASS], plast[PART2_MAXPASS];
    // dynamic LDS: the staging area of the final scatter (stage_cap entries, then one pass id per bucket); the planning phase's split-bucket lists live in
    // its first 3 x 2^PART2_MAXLOW words (dead before the scatter starts)
    extern __shared__ u32 stage[];
    u32* const xl_key = stage; u32* const xl_nt = stage + (1u << PART2_MAXLOW); u32* const xl_base = stage + (2u << PART2_MAXLOW);
    const u32 pidx = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const size_t q = (size_t)j * PART_P + pidx;
    const u32 nparts = gridDim.x * gridDim.y;
    const u32 nbl = 1u << low;
    for (u32 i = tid; i <= nblk; i += PART_T) pstart[i] = PO[q * nblk + i];           // (the scan has one element more than the matrix: the total)
    for (u32 i = tid; i < nblk; i += PART_T) {                                        // (window, point) of the first digit of pass-A block i
        const u32 est = (u32)((u64)tot_e * i / nblk), w0 = est / n;
        bw0[i] = w0; bi0[i] = est - w0 * n;
    }
    for (u32 i = tid; i < nbl; i += PART_T) cur[i] = 0;
    for (u32 i = tid; i < MAX_K + 2; i += PART_T) lh[i] = 0;
    if (tid == 0) xl_n = 0;
    __syncthreads();
    const u32 base = pstart[0], end = pstart[nblk];
    for (u32 x0 = base + tid; x0 < end; x0 += 8 * PART_T) {
        u32 m[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 x = x0 + u * PART_T; m[u] = mid[x < end ? x : end - 1]; }
#pragma unroll
        for (int u = 0; u < 8; u++) (void)lds_inc_agg(cur, m[u] >> 17, x0 + u * PART_T < end);
    }
    __syncthreads();
    {   // this thread's two consecutive buckets: entry offsets (off, the scatter cursors), task counts, their prefix, the length ranking
        const u32 i0 = 2 * tid, a = i0 < nbl ? cur[i0] : 0u, b = i0 + 1 < nbl ? cur[i0 + 1] : 0u;
        u32 btot;
        const u32 ex = base + block_scan_1024(a + b, sh, &btot);
        __syncthreads();
        const size_t key0 = (size_t)j * nb + (size_t)pidx * nbl + i0;
        if (i0 < nbl) { off[key0] = ex; cur[i0] = ex; }
        if (i0 + 1 < nbl) { off[key0 + 1] = ex + a; cur[i0 + 1] = ex + a; }
        if (tid == 0 && pidx == PART_P - 1 && j == last_group) off[(size_t)(j + 1) * nb] = end;          // off[nkeys] = number of entries
        const u32 K = pick_K(PO[total_idx], ta.room, ta.kmin, ta.ktab, ta.nkeys);
        const u32 nta = (a + K - 1) / K, ntb = (b + K - 1) / K;
        const u32 la = nta ? (a + nta - 1) / nta : 0u, lb = ntb ? (b + ntb - 1) / ntb : 0u;
        const u32 tex = block_scan_1024(nta + ntb, sh, &btot);
        if (i0 < nbl) toff[key0] = tex;
        if (i0 + 1 < nbl) toff[key0 + 1] = tex + nta;
        if (tid == 0) ptot[q] = btot;
        if (i0 < nbl) atomicAdd(&lh[la], 1u);
        if (i0 + 1 < nbl) atomicAdd(&lh[lb], 1u);
        if (nta > 1) { const u32 sl = atomicAdd(&xl_n, 1u); xl_key[sl] = (u32)key0; xl_nt[sl] = nta; xl_base[sl] = atomicAdd(&xlist[0], nta - 1u); slist[atomicAdd(&xlist[1], 1u)] = (u32)key0; }
        if (ntb > 1) { const u32 sl = atomicAdd(&xl_n, 1u); xl_key[sl] = (u32)key0 + 1u; xl_nt[sl] = ntb; xl_base[sl] = atomicAdd(&xlist[0], ntb - 1u); slist[atomicAdd(&xlist[1], 1u)] = (u32)key0 + 1u; }
        // hot buckets (more partials than one thread should add up): big[0] = their number, big[2 + i] = key, big[2 + bigcap + i] = the arrival counter of k_bucket_sum_wide
        if (nta > hot_nt) { const u32 sl = atomicAdd(&big[0], 1u); big[2 + sl] = (u32)key0; big[2 + bigcap + sl] = 0u; }
        if (ntb > hot_nt) { const u32 sl = atomicAdd(&big[0], 1u); big[2 + sl] = (u32)key0 + 1u; big[2 + bigcap + sl] = 0u; }
        // an empty bucket's record is the identity (true bucket number = local index << 8 | partition)
        if (i0 < nbl && a == 0) { uint4* rec = (uint4*)(buckets29 + ((size_t)j * nb + ((size_t)i0 << 8 | pidx)) * B29_BYTES); _Pragma("unroll") for (int t = 0; t < 9; t++) rec[t] = make_uint4(0u, 0u, 0u, 0u); }
        if (i0 + 1 < nbl && b == 0) { uint4* rec = (uint4*)(buckets29 + ((size_t)j * nb + ((size_t)(i0 + 1) << 8 | pidx)) * B29_BYTES); _Pragma("unroll") for (int t = 0; t < 9; t++) rec[t] = make_uint4(0u, 0u, 0u, 0u); }
        __syncthreads();
        // lh[L] <- number of buckets with a task length above L = first rank of length L (longest first)
        const u32 L = MAX_K - (tid <= MAX_K ? tid : MAX_K), hv = tid <= MAX_K ? lh[L] : 0u;
        const u32 above = block_scan_1024(hv, sh, &btot);
        __syncthreads();
        if (tid <= MAX_K) lh[L] = above;
        __syncthreads();
        if (i0 < nbl) { const u32 r = atomicAdd(&lh[la], 1u); order[(size_t)(r / og) * nparts * og + q * og + (r % og)] = (u32)key0; }
        if (i0 + 1 < nbl) { const u32 r = atomicAdd(&lh[lb], 1u); order[(size_t)(r / og) * nparts * og + q * og + (r % og)] = (u32)key0 + 1u; }
        // the chunks 1 .. nt - 1 of the split buckets, written by the whole block (one bucket may hold every entry of the MSM)
        const u32 nx = xl_n;
        for (u32 it = 0; it < nx; it++) {
            const u32 key = xl_key[it], ntx = xl_nt[it], xb = xl_base[it];
            for (u32 jt = 1 + tid; jt < ntx; jt += PART_T) { xlist[2 + 2 * (size_t)(xb + jt - 1)] = key; xlist[3 + 2 * (size_t)(xb + jt - 1)] = jt; }
        }
    }
    __syncthreads();
    const u32 pb0 = (u32)(pt_offset + (size_t)j * pt_batch);
    // The partition's records are the concatenation of the pass-A blocks' runs (~n W / (256 nblk) records each): a WAVE takes whole runs, so the
    // block that wrote a record -- and with it (window, point) of the record's position -- is wave-uniform (a per-record search through pstart cost
    // ten LDS reads per record).  Four 64-record chunks of a run are loaded before any is scattered.
    //
    // Where the entries go (round 6).  A partition's ~53 K entries cover 212 KB and 256 partitions are in flight: a 4-byte store to a random place in that
    // range leaves the L2 as a partial line long before its neighbours arrive -- 391 MB written for 54.5 MB of entries (profiles/r05_msm20_summary.txt), and
    // the kernel without these stores took 109 instead of 226 us in its phase (KH_DEBUG_PART2_NOSTORE, round 6).  So the scatter runs in PASSES over
    // consecutive bucket ranges of at most stage_cap - 1024 entries each: a pass scatters its buckets' entries into LDS and writes them out as one coalesced
    // run.  Every pass re-reads the partition's records (coalesced, L2-resident).  A bucket that straddles the staging area's end (longer than the 1024
    // margin: skew) has its tail stored directly.  Sweep (profiles/r06_part2_stage_sweep.txt, _sizes.txt; sort phase / pipelined rate, direct -> staged): 2^20 with
    // 14336 / 20480 / 28672 entries (4 / 3 / 2 passes) 231 -> 249 / 196 / 166 us, 1004 -> 984 / 1018 / 1035 Mscalar/s; 2^19 (one pass) 118 -> 83 us, 926 -> 961;
    // 2^21 (four passes) 491 -> 456 us but 1049 -> 1031; 2^22 (eight) 1112 -> 1398 us: every pass re-reads and re-filters the partition's records, so more than
    // max_pass passes (two by default: KH_PART2_MAXPASS) fall back to the direct scatter.
    const u32 lane = tid & 63u, wv = tid >> 6;
    const u32 S_eff = stage_cap > 2048u ? stage_cap - 1024u : 0u;
    const u32 npass = S_eff ? (end - base + S_eff - 1u) / S_eff : 0u;
    const bool staged = S_eff != 0u && npass >= 1u && npass <= max_pass;
    uint8_t* const bpass = (uint8_t*)(stage + stage_cap);
    if (staged) {
        if (tid < PART2_MAXPASS) { pfirst[tid] = 0xffffffffu; plast[tid] = 0u; }
        __syncthreads();
        for (u32 b = tid; b < nbl; b += PART_T) {          // cur[b] = the bucket's first entry position (set above)
            const u32 o = cur[b], oe = b + 1 < nbl ? cur[b + 1] : end, pb = (o - base) / S_eff;
            bpass[b] = (uint8_t)pb;
            if (oe > o) { atomicMin(&pfirst[pb], o); atomicMax(&plast[pb], oe); }
        }
        __syncthreads();
    }
    for (u32 pass = 0; pass < (staged ? npass : 1u); pass++) {
        const u32 p0 = staged ? pfirst[pass] : 0u, p1 = staged ? plast[pass] : 0u;
        if (staged && p0 == 0xffffffffu) continue;         // (no bucket starts in this slice: block-uniform)
        for (u32 blk = wv; blk < nblk; blk += PART_T / 64) {
            const u32 rs = pstart[blk], re = pstart[blk + 1];
            const u32 w0 = bw0[blk], i0 = bi0[blk];
            for (u32 x0 = rs; x0 < re; x0 += 256) {
                u32 m[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const u32 x = x0 + u * 64 + lane; m[u] = mid[x < re ? x : re - 1]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (x0 + u * 64 >= re) break;          // (wave-uniform)
                    const u32 bk = m[u] >> 17;
                    const bool live = x0 + u * 64 + lane < re && (!staged || bpass[bk] == pass);
                    const u32 pos = lds_inc_agg(cur, bk, live);
                    if (live) {
                        u32 w = w0, i = i0 + (m[u] & 0xffffu);
                        while (i >= n) { i -= n; w++; }
                        const u32 val = (pb0 + (u32)(w * pt_stride) + i) | ((m[u] & (1u << 16)) << 15);
                        if (staged && pos - p0 < stage_cap) stage[pos - p0] = val;
                        else entries[pos] = val;
                    }
                }
            }
        }
        if (staged) {
            __syncthreads();
            const u32 cnt = p1 - p0 < stage_cap ? p1 - p0 : stage_cap;
            for (u32 k = tid; k < cnt; k += PART_T) entries[p0 + k] = stage[k];
            __syncthreads();
        }
    }
}
// toff of k_part2_sort is relative to its partition: add the partitions' task bases; toff[nkeys] = number of tasks.  Block = 256 consecutive keys of
// ONE partition (2^low is a multiple of 256).  Also empties the hand-over and hot-bucket lists of the kernels that follow.
__global__ void __launch_bounds__(256)
k_wide_fixup(u32* __restrict__ toff, const u32* __restrict__ ptot, u32 nparts, u32 low, u32 nkeys, u32* __restrict__ handed, u32* __restrict__ big) {
    KH_HIGH_PRIO();
    __shared__ u32 sh[4];
    const u32 key = blockIdx.x * 256 + threadIdx.x, q = (blockIdx.x * 256) >> low;       // q == nparts for the block of key == nkeys
    u32 v = 0;
    for (u32 t = threadIdx.x; t < q && t < nparts; t += 256) v += ptot[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const u32 bs = sh[0] + sh[1] + sh[2] + sh[3];
    if (key < nkeys) toff[key] += bs;
    else if (key == nkeys) toff[key] = bs;
    if (key == 0) handed[0] = 0;                           // (big[0], the hot-bucket count, belongs to pass B on this path)
}
// ------------------------------------------------------------------------------------ 5 accumulate
template <class BF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(112)))
k_accumulate(const u32* __restrict__ entries, const u32* __restrict__ off, const u32* __restrict__ toff,
             const u32* __restrict__ roff, const u32* __restrict__ order,
             size_t nkeys, const uint8_t* __restrict__ pts, uint8_t* __restrict__ partial, const u32* __restrict__ only, const u32* __restrict__ abort_dev) {
    KH_HIGH_PRIO();
    if (sort_gave_up(abort_dev)) return;
  // `only` = the hand-over list of k_accumulate29 ([0] = count, then task ids): a small persistent grid walks it (almost always
  // empty -- a full-size grid of early-exit blocks took 0.6 ms to drain underneath the next job's accumulation)
  const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = only ? (size_t)gridDim.x * blockDim.x : 0;
  const size_t count = only ? only[0] : 1;
  for (size_t it = only ? first : 0; it < count; it += stride ? stride : 1) {
    size_t t = only ? only[1 + it] : first;
    u32 NT = roff[nkeys];
    if (t >= NT) return;
    // binary search over the length-ranked keys: largest rank with roff[rank] <= t
    size_t lo = 0, hi = nkeys;           // invariant roff[lo] <= t < roff[hi]
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (roff[mid] <= (u32)t) lo = mid; else hi = mid;
    }
    size_t key = order ? order[lo] : lo;
    u32 jt = (u32)t - roff[lo];
    t = (size_t)toff[key] + jt;          // the partial's slot stays in key order (k_bucket_sum)
    // balanced split of the bucket's entries over its nt = ceil(cnt / K) tasks: the lanes of a wave
    // run (almost) equally long chains instead of nt-1 full tasks + a short remainder
    u32 o0 = off[key], cnt = off[key + 1] - o0, nt = toff[key + 1] - toff[key];
    u32 start = o0 + (u32)(((u64)jt * cnt) / nt);
    u32 end = o0 + (u32)(((u64)(jt + 1) * cnt) / nt);
    u32 e = entries[start];
    Aff<BF> p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
    if (e >> 31) p.y = neg<BF>(p.y);
    Xyzz<BF> acc = Xyzz<BF>::from_affine(p);
    for (u32 k = start + 1; k < end; k++) {
        e = entries[k];
        p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
        acc = madd<BF>(acc, p, (e >> 31) != 0);
    }
    acc.store(partial + t * 128);
  }
}
// The same task in the lazy 29-bit-limb arithmetic of field29.cuh (186 instead of 254 instructions per product, no
// carry chains in the subtractions).  Its values are not canonical, so it cannot decide the exceptional cases of the
// group law; it only notices that one cannot be excluded (probability ~2^-25 per addition on random inputs), abandons
// the lazy state and redoes the task on the spot with the exact formulas of curve.cuh.  Finished
// tasks are converted to the canonical wire form, so everything downstream is unchanged and the result stays bit-exact.
// Occupancy is CAPPED at 4 waves per SIMD although 92 VGPRs would allow 5: the sort kernels of the next job (k_hist / k_scatter:
// 1024-thread blocks = 4 waves per SIMD, 128 KB of LDS) must find four free wave slots on every CU to run underneath this kernel;
// with 5 resident accumulate waves on some CUs they queued instead and an accumulate kernel was running only 73 % of the pipelined
// steady state (tools/timeline.py).
template <class BF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(112), amdgpu_waves_per_eu(4, 4)))
k_accumulate29(const u32* __restrict__ entries, const u32* __restrict__ off, const u32* __restrict__ toff,
               const u32* __restrict__ roff, const u32* __restrict__ order,
               size_t nkeys, const uint8_t* __restrict__ pts, uint8_t* __restrict__ partial, u32* __restrict__ handed, const u32* __restrict__ abort_dev, u32 hi_prio) {
    // MSM_LATENCY: the job is somebody's critical path (an opening round) and throughput work of the SAME process may be running underneath it at the
    // default priority (the rebase's side stream: a precompute wave sharing the SIMD is older and won every issue slot -- 90 -> 590 us, round 6 trace)
    if (hi_prio) __builtin_amdgcn_s_setprio(2);
    if (sort_gave_up(abort_dev)) return;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 NT = roff[nkeys];
    if (t0 >= NT) return;
    size_t lo = 0, hi = nkeys;           // invariant roff[lo] <= t0 < roff[hi]
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (roff[mid] <= (u32)t0) lo = mid; else hi = mid;
    }
    size_t key = order ? order[lo] : lo;
    u32 jt = (u32)t0 - roff[lo];
    const size_t t = (size_t)toff[key] + jt;
    u32 o0 = off[key], cnt = off[key + 1] - o0, nt = toff[key + 1] - toff[key];
    u32 start = o0 + (u32)(((u64)jt * cnt) / nt);
    u32 end = o0 + (u32)(((u64)(jt + 1) * cnt) / nt);
    u32 e = entries[start];
    Aff<BF> p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
    if (e >> 31) p.y = neg<BF>(p.y);
    typedef typename C29<BF>::T K29;
    Acc29<BF> acc;
    acc.x = to29<BF>(p.x); acc.y = to29<BF>(p.y);
#pragma unroll
    for (int i = 0; i < 9; i++) { acc.zz.v[i] = K29::one(i); acc.zzz.v[i] = K29::one(i); }
    bool ok = true;
    // (issuing the next point's gather before the current addition -- 109 VGPRs -- and sizing the tasks for 5 waves per SIMD
    //  were both measured: no gain, four waves already hide the gather)
    for (u32 k = start + 1; k < end; k++) {
        e = entries[k];
        p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
        if (e >> 31) p.y = neg<BF>(p.y);
        ok = madd29<BF>(acc, pack29<BF, 5>(p.x), pack29<BF, 5>(p.y));
        if (!ok) break;
    }
    Xyzz<BF> r;
    if (__builtin_expect(!ok, 0)) {
        // An exceptional case could not be excluded (~2^-25 per addition on random inputs; every time on degenerate bases): THIS lane redoes its task
        // with the exact formulas while the rest of its wave waits.  (Until round 5 such tasks went to a list for a second kernel: an empty launch
        // in every MSM -- 5 us in each of the 32 MSMs of an opening.)
        e = entries[start];
        p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
        if (e >> 31) p.y = neg<BF>(p.y);
        r = Xyzz<BF>::from_affine(p);
        for (u32 k = start + 1; k < end; k++) {
            e = entries[k];
            p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
            r = madd<BF>(r, p, (e >> 31) != 0);
        }
    } else {
        r.x = from29<BF>(acc.x); r.y = from29<BF>(acc.y); r.zz = from29<BF>(acc.zz); r.zzz = from29<BF>(acc.zzz);
    }
    r.store(partial + t * 128);
}
// ------------------------------------------------------------------------------------ 5w accumulate, wide windows
// One thread per BUCKET, no task search: thread g of the launch takes the bucket order[g] of k_part2_sort's interleaved length ranking (the longest
// bucket of every partition, then the second longest of every partition, ...: the 64 lanes of a wave run chains of nearly one length, and the launch
// as a whole runs longest-first -- with two rounds of resident blocks that order is worth 30 % of the kernel).  The bucket's only task writes the bucket
// itself, as a lazy B29 record (field29.cuh), into buckets29[key]: the two-plane reduction reads those.  A bucket above K entries (skewed scalars) is split
// as on the narrow path: thread g runs its first chunk, the others are listed by k_part2_sort as (key, chunk) pairs for k_acc_wide_rest; split
// buckets write wire-form partials for k_bucket_sum_wide.  Tasks whose lazy arithmetic cannot exclude an exceptional case are listed as (key, chunk)
// pairs too and redone exactly by k_acc_wide_rest.
// permuted key of the wide sort -> index of the bucket's B29 record (group * nb + true bucket number)
__device__ __forceinline__ u32 wide_true_bucket(u32 key, u32 low) {
    const u32 nbl = 1u << low, grp = key >> (low + 8), kk = key & ((nbl << 8) - 1u);
    return (grp << (low + 8)) | ((kk & (nbl - 1u)) << 8) | (kk >> low);
}
// (returns false only when the lazy arithmetic gave up AND there is no hand-over list: the caller redoes the task itself)
template <class BF>
__device__ __forceinline__ bool wide_task29(u32 key, u32 jt, const u32* __restrict__ entries, const u32* __restrict__ off, const u32* __restrict__ toff,
                                            const uint8_t* __restrict__ pts, uint8_t* __restrict__ partial, uint8_t* __restrict__ buckets29, u32 low, u32* __restrict__ handed) {
    const u32 o0 = off[key], cnt = off[key + 1] - o0;
    if (cnt == 0) return true;
    const u32 t0 = toff[key], nt = toff[key + 1] - t0;
    const u32 start = o0 + (u32)(((u64)jt * cnt) / nt), end = o0 + (u32)(((u64)(jt + 1) * cnt) / nt);
    u32 e = entries[start];
    Aff<BF> p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
    if (e >> 31) p.y = neg<BF>(p.y);
    typedef typename C29<BF>::T K29;
    Acc29<BF> acc;
    acc.x = to29<BF>(p.x); acc.y = to29<BF>(p.y);
#pragma unroll
    for (int i = 0; i < 9; i++) { acc.zz.v[i] = K29::one(i); acc.zzz.v[i] = K29::one(i); }
    bool ok = true;
    for (u32 k = start + 1; k < end; k++) {
        e = entries[k];
        p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
        if (e >> 31) p.y = neg<BF>(p.y);
        ok = madd29<BF>(acc, pack29<BF, 5>(p.x), pack29<BF, 5>(p.y));
        if (!ok) break;
    }
    // (the exact redo stays a separate kernel here: inlined as in k_accumulate29 it raised this kernel from 92 to 127 VGPRs, and the sort kernels of the
    //  neighbouring job no longer fitted beside three resident blocks: 1020 -> 985 Mscalar/s pipelined, profiles/r05_ab_inline_exact.txt)
    if (!ok) { if (!handed) return false; const u32 slot = atomicAdd(&handed[0], 1u); handed[2 + 2 * slot] = key; handed[3 + 2 * slot] = jt; return true; }
    if (nt == 1) { store_b29<BF>(buckets29 + (size_t)wide_true_bucket(key, low) * B29_BYTES, acc); return true; }
    Xyzz<BF> r;
    r.x = from29<BF>(acc.x); r.y = from29<BF>(acc.y); r.zz = from29<BF>(acc.zz); r.zzz = from29<BF>(acc.zzz);
    r.store(partial + (size_t)(t0 + jt) * 128);
    return true;
}
// the same task with the exact formulas (complete addition: equal points, opposite points, the identity)
template <class BF>
__device__ __forceinline__ void wide_task_exact(u32 key, u32 jt, const u32* __restrict__ entries, const u32* __restrict__ off, const u32* __restrict__ toff,
                                                const uint8_t* __restrict__ pts, uint8_t* __restrict__ partial, uint8_t* __restrict__ buckets29, u32 low) {
    const u32 o0 = off[key], cnt = off[key + 1] - o0, t0 = toff[key], nt = toff[key + 1] - t0;
    if (cnt == 0) return;
    const u32 start = o0 + (u32)(((u64)jt * cnt) / nt), end = o0 + (u32)(((u64)(jt + 1) * cnt) / nt);
    u32 e = entries[start];
    Aff<BF> p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
    if (e >> 31) p.y = neg<BF>(p.y);
    Xyzz<BF> acc = Xyzz<BF>::from_affine(p);
    for (u32 k = start + 1; k < end; k++) {
        e = entries[k];
        p = Aff<BF>::load(pts + (size_t)(e & 0x7fffffffu) * 64);
        acc = madd<BF>(acc, p, (e >> 31) != 0);
    }
    if (nt == 1) {
        uint8_t* const rec = buckets29 + (size_t)wide_true_bucket(key, low) * B29_BYTES;
        if (acc.is_identity()) store_b29_identity<BF>(rec); else store_b29<BF>(rec, xyzz_to29<BF>(acc));
    } else acc.store(partial + (size_t)(t0 + jt) * 128);
}
// Thread per bucket.  Launched with enough dynamic LDS to hold the kernel to THREE blocks per CU (it is VALU-bound from
// three waves per SIMD on; the fourth only keeps the neighbouring jobs' sort and reduction kernels off the CU: 886-930 -> 940-1000 Mscalar/s pipelined).
static constexpr u32 WIDE_XB = 32, WIDE_HOT_NT = 16;       // (a split bucket with more partials than WIDE_HOT_NT is summed by many waves)
template <class BF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(112), amdgpu_waves_per_eu(4, 4)))
k_acc_wide29(const u32* __restrict__ entries, const u32* __restrict__ off, const u32* __restrict__ toff, const u32* __restrict__ order, u32 nkeys,
             const uint8_t* __restrict__ pts, uint8_t* __restrict__ partial, uint8_t* __restrict__ buckets29, u32 low, u32* __restrict__ handed) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < nkeys) (void)wide_task29<BF>(order[g], 0u, entries, off, toff, pts, partial, buckets29, low, handed);
}
// Everything k_acc_wide29 left behind, in ONE launch (round 6; until then two: the extra chunks, then the exact redos -- each an empty dependent launch for
// unskewed scalars): a small persistent grid walks (a) xlist, the extra chunks of split buckets ([0] = their number, then (key, chunk) pairs from word 2 on;
// empty for unskewed scalars) -- in the lazy arithmetic, and a chunk that gives up is redone exactly by the same thread, nothing is appended to `handed`
// here --, then (b) `handed`, the tasks k_acc_wide29 gave up on (complete before this kernel starts).  A kernel of its own: walked by blocks of k_acc_wide29
// the loop cost that kernel 13 VGPRs (92 -> 105) and the inline exact redo 35 more, and with three resident blocks per CU the neighbouring job's 1024-thread
// sort kernels (4 x 56 VGPRs per SIMD) no longer fitted beside them (1020 -> 985 Mscalar/s pipelined, profiles/r05_ab_inline_exact.txt).
template <class BF>
__global__ void __launch_bounds__(256)
k_acc_wide_rest(const u32* __restrict__ entries, const u32* __restrict__ off, const u32* __restrict__ toff, const u32* __restrict__ xlist, const u32* __restrict__ handed,
                const uint8_t* __restrict__ pts, uint8_t* __restrict__ partial, uint8_t* __restrict__ buckets29, u32 low) {
    KH_HIGH_PRIO();
    const u32 nx = xlist[0], nh = handed[0];
    for (u32 it = blockIdx.x * blockDim.x + threadIdx.x; it < nx; it += gridDim.x * blockDim.x) {
        const u32 key = xlist[2 + 2 * it], jt = xlist[3 + 2 * it];
        if (!wide_task29<BF>(key, jt, entries, off, toff, pts, partial, buckets29, low, nullptr))
            wide_task_exact<BF>(key, jt, entries, off, toff, pts, partial, buckets29, low);
    }
    for (u32 it = blockIdx.x * blockDim.x + threadIdx.x; it < nh; it += gridDim.x * blockDim.x)
        wide_task_exact<BF>(handed[2 + 2 * it], handed[3 + 2 * it], entries, off, toff, pts, partial, buckets29, low);
}
// ------------------------------------------------------------------------------------ 6 bucket sums
// buckets with more partials than this go to the wave-per-bucket tree (k_bucket_big); the threshold is a
// kernel argument: 16 for large problems (throughput), 4 for small ones (shorter dependent chain)
// hot-bucket bookkeeping in `big`: [0] = #big buckets, [1] = #chunk items, then four arrays of `cap` words:
// bigkey[], bigcbase[] (first chunk item of that bucket), itemkey[], itemj[]
static constexpr u32 CHUNK = 128;            // partials per chunk item = 8 per quad of a wave (cooperative additions)
template <class BF>
__global__ void __launch_bounds__(256)
k_bucket_sum(const u32* __restrict__ toff, size_t nkeys, const uint8_t* __restrict__ partial,
             uint8_t* __restrict__ buckets, u32* __restrict__ big, size_t cap, u32 SMALL_NT, const u32* __restrict__ abort_dev) {
    KH_HIGH_PRIO();
    if (sort_gave_up(abort_dev)) return;
    size_t key = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    u32 t0 = toff[key], nt = toff[key + 1] - t0;
    Xyzz<BF> acc = Xyzz<BF>::identity();
    if (nt > SMALL_NT) {
        u32 nch = (nt + CHUNK - 1) / CHUNK;
        u32 slot = atomicAdd(&big[0], 1u);
        u32 cbase = atomicAdd(&big[1], nch);
        big[2 + slot] = (u32)key; big[2 + cap + slot] = cbase;
        for (u32 j = 0; j < nch; j++) { big[2 + 2 * cap + cbase + j] = (u32)key; big[2 + 3 * cap + cbase + j] = j; }
    } else if (nt > 0) {
        acc = Xyzz<BF>::load(partial + (size_t)t0 * 128);
        for (u32 k = 1; k < nt; k++) acc = add<BF>(acc, Xyzz<BF>::load(partial + (size_t)(t0 + k) * 128));
    }
    acc.store(buckets + key * 128);
}
template <class F>
__device__ __forceinline__ Xyzz<F> shfl_down(const Xyzz<F>& a, int delta) {
    Xyzz<F> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.v[i] = __shfl_down(a.x.v[i], delta, 64); r.y.v[i] = __shfl_down(a.y.v[i], delta, 64);
        r.zz.v[i] = __shfl_down(a.zz.v[i], delta, 64); r.zzz.v[i] = __shfl_down(a.zzz.v[i], delta, 64);
    }
    return r;
}
// Hot buckets in two phases, so that one bucket holding most of the entries (the all-ones witness) is
// summed by many waves: phase A = one wave per chunk of <= 512 partials (8 per lane + shuffle tree),
// phase B = one wave per hot bucket over its chunk sums.
template <class BF>
__global__ void __launch_bounds__(64)
k_bucket_chunk(const u32* __restrict__ toff, const uint8_t* __restrict__ partial, const u32* __restrict__ big, size_t cap,
               uint8_t* __restrict__ chunk_out, const u32* __restrict__ abort_dev) {
    KH_HIGH_PRIO();
    if (sort_gave_up(abort_dev)) return;
    // one wave = 16 quads per chunk item; every addition is the lane-cooperative one (coop.cuh): 8 sequential + 4 tree levels
    u32 nitems = big[1];
    const u32 quad = threadIdx.x >> 2;
    for (u32 it = blockIdx.x; it < nitems; it += gridDim.x) {
        u32 key = big[2 + 2 * cap + it], j = big[2 + 3 * cap + it];
        u32 t0 = toff[key], nt = toff[key + 1] - t0;
        u32 lo = j * CHUNK, hi = lo + CHUNK < nt ? lo + CHUNK : nt;
        Fe<BF> acc = Fe<BF>::zero();
        for (u32 k = lo + quad; k < hi; k += 16) acc = quad_add<BF>(acc, quad_load<BF>(partial + (size_t)(t0 + k) * 128));
        for (int d = 8; d >= 1; d >>= 1) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));
        if (threadIdx.x < 4) quad_store<BF>(chunk_out + (size_t)it * 128, acc);
    }
}
template <class BF>
__global__ void __launch_bounds__(256)
k_bucket_big(const u32* __restrict__ toff, const uint8_t* __restrict__ chunk_out, uint8_t* __restrict__ buckets,
             const u32* __restrict__ big, size_t cap, const u32* __restrict__ abort_dev) {
    KH_HIGH_PRIO();
    if (sort_gave_up(abort_dev)) return;
    // one 256-thread block = 64 quads per hot bucket: a bucket holding 2^20 entries has ~1000 chunk sums
    __shared__ u32 sh[4 * 32];
    u32 nbig = big[0];
    const u32 quad = threadIdx.x >> 2, role = threadIdx.x & 3u, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (u32 bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        u32 key = big[2 + bi], cbase = big[2 + cap + bi];
        u32 nt = toff[key + 1] - toff[key];
        u32 nch = (nt + CHUNK - 1) / CHUNK;
        Fe<BF> acc = Fe<BF>::zero();
        for (u32 k = quad; k < nch; k += 64) acc = quad_add<BF>(acc, quad_load<BF>(chunk_out + (size_t)(cbase + k) * 128));
        for (int d = 8; d >= 1; d >>= 1) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));
        __syncthreads();                                    // sh is reused across iterations
        if (lane < 4) {
#pragma unroll
            for (int k = 0; k < 8; k++) sh[(wave * 4 + role) * 8 + k] = acc.v[k];
        }
        __syncthreads();
        if (wave == 0) {
            Fe<BF> o = Fe<BF>::zero();
            if (quad < 4) {
#pragma unroll
                for (int k = 0; k < 8; k++) o.v[k] = sh[(quad * 4 + role) * 8 + k];
            }
            acc = o;
            for (int d = 2; d >= 1; d >>= 1) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));
            if (threadIdx.x < 4) quad_store<BF>(buckets + (size_t)key * 128, acc);
        }
    }
}
// ------------------------------------------------------------------------------------ 7 reduce
// thread (q, t): segment of m buckets [t*m, (t+1)*m) of group q -> sum_b (b+1) B_b restricted to it
template <class BF>
__global__ void __launch_bounds__(128)
k_reduce_seg(const uint8_t* __restrict__ buckets, u32 nb, u32 m, size_t ngroups, uint8_t* __restrict__ seg) {
    KH_HIGH_PRIO();
    u32 nseg = nb / m;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= ngroups * nseg) return;
    size_t q = gid / nseg; u32 t = (u32)(gid % nseg);
    const uint8_t* B = buckets + (q * nb + (size_t)t * m) * 128;
    Xyzz<BF> run = Xyzz<BF>::identity(), acc = Xyzz<BF>::identity();
    for (int b = (int)m - 1; b >= 0; b--) {
        run = add<BF>(run, Xyzz<BF>::load(B + (size_t)b * 128));
        acc = add<BF>(acc, run);
    }
    // + (t*m) * run   (double-and-add, t*m < nb <= 2^15)
    u32 lo = t * m;
    if (lo) {
        Xyzz<BF> r = Xyzz<BF>::identity();
        for (int bit = 31 - __clz(lo); bit >= 0; bit--) {
            r = dbl<BF>(r);
            if ((lo >> bit) & 1u) r = add<BF>(r, run);
        }
        acc = add<BF>(acc, r);
    }
    acc.store(seg + gid * 128);
}
// sum of `count` XYZZ records per group, two launches: level 1 = blocks of 256 threads x SUM_PER_T
// records (coalesced strided loads, shuffle tree, LDS across the 4 waves), level 2 = one block per group.
static constexpr u32 SUM_PER_T = 2, SUM_BLK = 256 * SUM_PER_T;
template <class BF>
__device__ __forceinline__ Xyzz<BF> block_sum(Xyzz<BF> acc, u32* sh) {
    for (int d = 32; d >= 1; d >>= 1) { Xyzz<BF> o = shfl_down<BF>(acc, d); acc = add<BF>(acc, o); }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32* mine = sh + wave * 32;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) { mine[i] = acc.x.v[i]; mine[8 + i] = acc.y.v[i]; mine[16 + i] = acc.zz.v[i]; mine[24 + i] = acc.zzz.v[i]; }
    }
    __syncthreads();
    int nw = blockDim.x >> 6;
    if (threadIdx.x < 64) {              // wave 0 folds the (<= 4) wave results
        Xyzz<BF> o = Xyzz<BF>::identity();
        if (lane < nw) {
            const u32* p = sh + lane * 32;
#pragma unroll
            for (int i = 0; i < 8; i++) { o.x.v[i] = p[i]; o.y.v[i] = p[8 + i]; o.zz.v[i] = p[16 + i]; o.zzz.v[i] = p[24 + i]; }
        }
        for (int d = 2; d >= 1; d >>= 1) { Xyzz<BF> q = shfl_down<BF>(o, d); o = add<BF>(o, q); }
        acc = o;
    }
    return acc;                          // valid in thread 0
}
template <class BF>
__global__ void __launch_bounds__(256)
k_sum_level(const uint8_t* __restrict__ in, u32 count, u32 out_stride, uint8_t* __restrict__ out) {
    KH_HIGH_PRIO();
    __shared__ u32 sh[4 * 32];
    size_t q = blockIdx.y;
    const uint8_t* S = in + q * (size_t)count * 128;
    u32 base = blockIdx.x * SUM_BLK;
    Xyzz<BF> acc = Xyzz<BF>::identity();
    for (u32 k = base + threadIdx.x; k < count && k < base + SUM_BLK; k += 256) acc = add<BF>(acc, Xyzz<BF>::load(S + (size_t)k * 128));
    acc = block_sum<BF>(acc, sh);
    if (threadIdx.x == 0) acc.store(out + (q * out_stride + blockIdx.x) * 128);
}
template <class BF>
__global__ void __launch_bounds__(256)
k_sum_final(const uint8_t* __restrict__ in, u32 count, uint8_t* __restrict__ out) {
    KH_HIGH_PRIO();
    __shared__ u32 sh[4 * 32];
    size_t q = blockIdx.x;
    const uint8_t* S = in + q * (size_t)count * 128;
    Xyzz<BF> acc = Xyzz<BF>::identity();
    for (u32 k = threadIdx.x; k < count; k += blockDim.x) acc = add<BF>(acc, Xyzz<BF>::load(S + (size_t)k * 128));
    acc = block_sum<BF>(acc, sh);
    if (threadIdx.x == 0) acc.store(out + q * 128);
}

// Latency path of the weighted reduction for one to eight MSMs over the tables (the per-round L / R of an opening, the seven chunks of t,
// a lone commitment).  Bucket t has weight t + 1; with t = (a, b, c) split into three digit fields,
//   sum_t (t+1) B_t = 2^(f0+f1) sum_a a A_a + 2^f0 sum_b b B_b + sum_c (c+1) C_c ,
// A_a / B_b / C_c = the marginal sums of the buckets over the other two digits.  The marginals are plain tree
// sums of nb / 32 buckets each (k_marginals: 4 sequential additions + the block tree), the digit weights then need
// 5-bit ladders on 3 x 32 points only (k_marginal_fin), and the host folds the three results with f0 + f1 doublings.
// Chain ~380 field products instead of ~630 (15-bit ladder per bucket + two tree levels), and a tenth of the work.
struct MargGeom { u32 nb; u32 sh[3]; u32 wd[3]; };
template <class BF>
__global__ void __launch_bounds__(256)
k_marginals(const uint8_t* __restrict__ buckets, MargGeom g, uint8_t* __restrict__ out) {
    KH_HIGH_PRIO();
    __shared__ u32 sh[4 * 32];
    const u32 v = blockIdx.x, j = blockIdx.y; const size_t q = blockIdx.z;
    const u32 s = g.sh[j], f = g.wd[j];
    if (v >= (1u << f)) return;
    const uint8_t* B = buckets + q * (size_t)g.nb * 128;
    const u32 cnt = g.nb >> f, lowmask = (1u << s) - 1u;
    Xyzz<BF> acc = Xyzz<BF>::identity();
    for (u32 e = threadIdx.x; e < cnt; e += blockDim.x) {
        const u32 t = ((e >> s) << (s + f)) | (v << s) | (e & lowmask);
        acc = add<BF>(acc, Xyzz<BF>::load(B + (size_t)t * 128));
    }
    acc = block_sum<BF>(acc, sh);
    if (threadIdx.x == 0) acc.store(out + ((q * 3 + j) * 32 + v) * 128);
}
// block (j, q): sum_v (v + [j == 0]) M_v over the <= 32 marginals of digit j, as the sum of the suffix sums
// S_k = sum_{v >= k} M_v over k >= 1 (k >= 0 for j == 0): a 5-step scan and a 5-step tree, no doublings
template <class BF>
__global__ void __launch_bounds__(64)
k_marginal_fin(const uint8_t* __restrict__ marg, MargGeom g, uint8_t* __restrict__ out) {
    KH_HIGH_PRIO();
    const u32 j = blockIdx.x; const size_t q = blockIdx.y;
    const u32 v = threadIdx.x, f = g.wd[j];
    Xyzz<BF> r = Xyzz<BF>::identity();
    if (v < (1u << f)) r = Xyzz<BF>::load(marg + ((q * 3 + j) * 32 + v) * 128);
    for (u32 d = 1; d < 32; d <<= 1) {                    // inclusive suffix scan over lanes 0..31 (lanes >= 32 hold the identity)
        Xyzz<BF> o = shfl_down<BF>(r, (int)d);
        if (v + d >= 64) o = Xyzz<BF>::identity();
        r = add<BF>(r, o);
    }
    if (v == 0 && j != 0) r = Xyzz<BF>::identity();       // weight of digit value 0
    if (v >= 32) r = Xyzz<BF>::identity();
    for (int d = 16; d >= 1; d >>= 1) { Xyzz<BF> o = shfl_down<BF>(r, d); r = add<BF>(r, o); }
    if (threadIdx.x == 0) r.store(out + (q * 3 + j) * 128);
}

// quad per bucket: the bucket's task partials summed with the cooperative addition (latency path)
template <class BF>
__global__ void __launch_bounds__(256)
k_bucket_sum_q(const u32* __restrict__ toff, size_t nkeys, const uint8_t* __restrict__ partial,
               uint8_t* __restrict__ buckets, u32* __restrict__ big, size_t cap, u32 SMALL_NT, const u32* __restrict__ abort_dev,
               u32* __restrict__ spread_abort) {
    KH_HIGH_PRIO();
    if (sort_gave_up(abort_dev)) return;
    size_t key = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const bool live = key < nkeys;
    if (!live) key = nkeys - 1;
    const u32 role = threadIdx.x & 3u;
    u32 t0 = toff[key], nt = toff[key + 1] - t0;
    Fe<BF> acc = Fe<BF>::zero();
    if (nt > SMALL_NT) {
        if (spread_abort) {                                // no hot-bucket kernels behind this launch (MSM_SPREAD_SCALARS): tell the host, which re-runs the job with them
            if (live && role == 0) __hip_atomic_store(spread_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else
        if (live && role == 0) {
            u32 nch = (nt + CHUNK - 1) / CHUNK;
            u32 slot = atomicAdd(&big[0], 1u);
            u32 cbase = atomicAdd(&big[1], nch);
            big[2 + slot] = (u32)key; big[2 + cap + slot] = cbase;
            for (u32 j = 0; j < nch; j++) { big[2 + 2 * cap + cbase + j] = (u32)key; big[2 + 3 * cap + cbase + j] = j; }
        }
        nt = 0;
    }
    // the trip count may differ between the quads of a wave: quad_add's shuffles stay inside the quad, its ballot is only a hint
    if (nt) acc = quad_load<BF>(partial + (size_t)t0 * 128);            // (not "identity + first": an addition costs 5 product rounds)
    for (u32 k = 1; k < nt; k++) acc = quad_add<BF>(acc, quad_load<BF>(partial + (size_t)(t0 + k) * 128));
    if (live) quad_store<BF>(buckets + key * 128, acc);
}

// Quad versions of the two reduction kernels for the latency path (<= 8 MSMs): the same marginal sums with the
// lane-cooperative addition of coop.cuh -- every addition of the chain costs 5 product rounds instead of 14 products.
template <class BF>
__global__ void __launch_bounds__(1024)
k_marginals_q(const uint8_t* __restrict__ buckets, MargGeom g, uint8_t* __restrict__ out) {
    KH_HIGH_PRIO();
    __shared__ u32 sh[16 * 32];
    const u32 v = blockIdx.x, j = blockIdx.y; const size_t q = blockIdx.z;
    const u32 s = g.sh[j], f = g.wd[j];
    if (v >= (1u << f)) return;
    const uint8_t* B = buckets + q * (size_t)g.nb * 128;
    const u32 cnt = g.nb >> f, lowmask = (1u << s) - 1u;
    const u32 quad = threadIdx.x >> 2, nquads = blockDim.x >> 2, role = threadIdx.x & 3u;
    Fe<BF> acc = Fe<BF>::zero();
    for (u32 e = quad; e < cnt; e += nquads) {
        const u32 t = ((e >> s) << (s + f)) | (v << s) | (e & lowmask);
        acc = quad_add<BF>(acc, quad_load<BF>(B + (size_t)t * 128));
    }
    for (int d = 8; d >= 1; d >>= 1) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));       // 16 quads of the wave
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 8; k++) sh[(wave * 4 + role) * 8 + k] = acc.v[k];
    }
    __syncthreads();
    if (wave == 0) {                                        // quad w of wave 0 takes wave w's sum
        Fe<BF> o = Fe<BF>::zero();
        if (quad < nw) {
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[(quad * 4 + role) * 8 + k];
        }
        acc = o;
        for (int d = 8; d >= 1; d >>= 1) if ((u32)d < nw) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));
        if (threadIdx.x < 4) quad_store<BF>(out + ((q * 3 + j) * 32 + v) * 128, acc);
    }
}
// Hybrid of the two: the sequential part with ONE LANE per addition, the tree with quads.  These tails are bound by VALU ISSUE, not only by the
// dependent chain (a cooperative addition is 5 product rounds x 4 lanes = 20 lane-products, the ordinary one 14 in one lane: per addition 78 against 55
// wave-instruction slots), so where every lane has its own bucket the ordinary addition is the cheaper one, and where the tree leaves lanes idle the
// cooperative one is the shorter one.  256 threads: each lane adds its cnt / 256 buckets (4 at 2^15 buckets), parks the sum in LDS (word-major: no bank
// conflicts), then each quad folds four parked points of its wave and the 16 quads / 4 waves finish as in k_marginals_q.
template <class BF>
__global__ void __launch_bounds__(256)
k_marginals_h(const uint8_t* __restrict__ buckets, MargGeom g, uint8_t* __restrict__ out) {
    KH_HIGH_PRIO();
    __shared__ u32 stage[4][32][64];                    // [wave][word of the XYZZ record][lane]
    __shared__ u32 sh[16 * 32];
    const u32 v = blockIdx.x, j = blockIdx.y; const size_t q = blockIdx.z;
    const u32 s = g.sh[j], f = g.wd[j];
    if (v >= (1u << f)) return;
    const uint8_t* B = buckets + q * (size_t)g.nb * 128;
    const u32 cnt = g.nb >> f, lowmask = (1u << s) - 1u;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, role = threadIdx.x & 3u, quad = lane >> 2, nw = blockDim.x >> 6;
    Xyzz<BF> a = Xyzz<BF>::identity();
    bool have = false;
    for (u32 e = threadIdx.x; e < cnt; e += blockDim.x) {
        const u32 t = ((e >> s) << (s + f)) | (v << s) | (e & lowmask);
        const Xyzz<BF> b = Xyzz<BF>::load(B + (size_t)t * 128);
        a = have ? add<BF>(a, b) : b;
        have = true;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { stage[wave][k][lane] = a.x.v[k]; stage[wave][8 + k][lane] = a.y.v[k]; stage[wave][16 + k][lane] = a.zz.v[k]; stage[wave][24 + k][lane] = a.zzz.v[k]; }
    __syncthreads();
    auto parked = [&](u32 point) {                       // this lane's coordinate (role) of a point parked by lane `point` of this wave
        Fe<BF> r;
#pragma unroll
        for (int k = 0; k < 8; k++) r.v[k] = stage[wave][role * 8 + k][point];
        return r;
    };
    Fe<BF> acc = parked(quad);
    for (u32 k = 1; k < 4; k++) acc = quad_add<BF>(acc, parked(quad + 16 * k));
    for (int d = 8; d >= 1; d >>= 1) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));       // 16 quads of the wave
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 8; k++) sh[(wave * 4 + role) * 8 + k] = acc.v[k];
    }
    __syncthreads();
    if (wave == 0) {                                        // quad w of wave 0 takes wave w's sum
        Fe<BF> o = Fe<BF>::zero();
        if (quad < nw) {
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[(quad * 4 + role) * 8 + k];
        }
        acc = o;
        for (int d = 8; d >= 1; d >>= 1) if ((u32)d < nw) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));
        if (threadIdx.x < 4) quad_store<BF>(out + ((q * 3 + j) * 32 + v) * 128, acc);
    }
}
// block (j, q), 128 threads = 32 quads, one per marginal: suffix scan + sum as in k_marginal_fin, exchanges through LDS
template <class BF>
__global__ void __launch_bounds__(128)
k_marginal_fin_q(const uint8_t* __restrict__ marg, MargGeom g, uint8_t* __restrict__ out, u32* __restrict__ done_ws, u32* __restrict__ done_flag) {
    KH_HIGH_PRIO();
    __shared__ u32 sh[32 * 32];
    const u32 j = blockIdx.x; const size_t q = blockIdx.y;
    const u32 v = threadIdx.x >> 2, role = threadIdx.x & 3u, f = g.wd[j];
    Fe<BF> r = Fe<BF>::zero();
    if (v < (1u << f)) r = quad_load<BF>(marg + ((q * 3 + j) * 32 + v) * 128);
    auto put = [&](const Fe<BF>& x) {
#pragma unroll
        for (int k = 0; k < 8; k++) sh[(v * 4 + role) * 8 + k] = x.v[k];
    };
    auto get = [&](u32 from) {
        Fe<BF> o = Fe<BF>::zero();
        if (from < 32) {
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[(from * 4 + role) * 8 + k];
        }
        return o;
    };
    for (u32 d = 1; d < 32; d <<= 1) {                     // inclusive suffix scan over the 32 quads
        put(r); __syncthreads();
        const Fe<BF> o = get(v + d);
        __syncthreads();
        r = quad_add<BF>(r, o);
    }
    if (v == 0 && j != 0) r = Fe<BF>::zero();              // weight of digit value 0
    for (u32 d = 16; d >= 1; d >>= 1) {
        put(r); __syncthreads();
        const Fe<BF> o = v < d ? get(v + d) : Fe<BF>::zero();
        __syncthreads();
        r = quad_add<BF>(r, o);
    }
    if (threadIdx.x < 4) quad_store<BF>(out + (q * 3 + j) * 128, r);
    if (done_ws) {                                         // completion by flag (common.hpp, MsmSlot::done_flag): `out` is pinned host memory
        __threadfence_system();                            // this thread's result words are visible to the host ...
        __syncthreads();
        if (threadIdx.x == 0) {                            // ... before the launch count is, which the LAST block to get here stores
            const u32 blocks = gridDim.x * gridDim.y;
            if (__hip_atomic_fetch_add(done_ws, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == blocks - 1) {
                __hip_atomic_store(done_ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-armed: launches on a slot are stream-ordered
                const u32 e = done_ws[1] + 1;
                done_ws[1] = e;
                __threadfence_system();
                __hip_atomic_store(done_flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ 6w / 7w wide windows: bucket sums + two-plane reduction
// c = 20 leaves 2^19 buckets of ~26 entries: 13 instead of 16 table additions per scalar, paid for by a reduction over 16 x more buckets.
// Bucket t = (a, b), a = its top `hi` bits, b = its low `lo` bits, has weight t + 1:
//     sum_t (t + 1) B_t  =  2^lo  sum_a a H_a  +  sum_b (b + 1) L_b,      H_a = sum_b B_(a,b),   L_b = sum_a B_(a,b)
// -- every bucket goes into ONE hi-digit and ONE lo-digit marginal: 2 x 2^19 full additions, all of them in the lazy 29-bit arithmetic
// (add29: ~1.4 mixed additions each), then 2^hi + 2^lo = 1536 marginals take the 5-bit digit-marginal tail of the narrow path.
//   k_bucket_sum_wide  the split buckets of pass B's list (exact sum of the wire partials; hot ones by the whole block, lane-cooperatively);
//                      empty buckets got their identity record from pass B, all others ARE a task's output
//   k_wide_a1          thread (plane, marginal, chunk): the sum of r = 8 buckets, sequentially, B29 in -> B29 out (2^17 threads; plane 0 reads
//                      r consecutive records, plane 1 a column: neighbouring lanes read neighbouring records).  A thread whose add29 cannot
//                      exclude an exceptional case writes a marker record; k_wide_a2 redoes such a chunk with the exact formulas.
//   k_wide_a2          wave per marginal: its 2^lo / r (or 2^hi / r) chunk sums, converted to the wire form, one or two per lane, then the
//                      lane-cooperative tree of k_marginals_h.  Output: 2 x 2^lo wire records per MSM laid out as TWO bucket groups of 2^lo
//                      buckets for k_marginals_q / k_marginal_fin_q -- group 0 holds H_a at slot a - 1 (weight a; H_0 has weight 0 and is
//                      dropped, slots >= 2^hi - 1 are the identity), group 1 holds L_b at slot b.
struct WideGeom { u32 nb, lo, hi, rlog; };
// split buckets only (slist: pass B's list of buckets with more than one task; count in xlist[1]), ONE launch (round 6; until then this kernel listed the hot
// buckets for k_bucket_chunk / k_bucket_big and k_big_to29 converted their sums: three more empty dependent launches for unskewed scalars).
//   (1) a thread sums a bucket of up to SMALL_NT partials by itself (2^22 uniform scalars: 16 K buckets of two partials);
//   (2) the hot buckets are on a list of their own (pass B wrote it: big[0] = count, big[2 + i] = key, big[2 + cap + i] = 0).  EVERY wave of the grid walks that
//       list and takes a strided share of every bucket's chunks of CHUNK partials (one bucket may hold all 2^20 entries of a window: 32 K partials at K = 32 --
//       a single block on it took 3.2 ms): 16 quads, lane-cooperative additions (coop.cuh), 8 sequential + 4 tree levels per chunk, the chunk's sum to
//       chunk_out.  The wave that brings a bucket's LAST chunk in (a counter per bucket, fenced on both sides) adds the chunk sums up and writes the B29 record:
//       no second kernel, no grid barrier.
template <class BF>
__global__ void __launch_bounds__(256)
k_bucket_sum_wide(const u32* __restrict__ toff, const u32* __restrict__ xlist, const u32* __restrict__ slist, const uint8_t* __restrict__ partial,
                  uint8_t* __restrict__ buckets29, u32 low, u32 SMALL_NT, u32* __restrict__ big, u32 cap, uint8_t* __restrict__ chunk_out) {
    KH_HIGH_PRIO();
    __shared__ __attribute__((aligned(16))) uint8_t total[4 * 128];
    const u32 count = xlist[1], nhot = big[0];
    for (u32 it = blockIdx.x * blockDim.x + threadIdx.x; it < count; it += gridDim.x * blockDim.x) {
        const u32 key = slist[it];
        const u32 t0 = toff[key], nt = toff[key + 1] - t0;
        if (nt > SMALL_NT) continue;
        Xyzz<BF> acc = Xyzz<BF>::load(partial + (size_t)t0 * 128);
        for (u32 k = 1; k < nt; k++) acc = add<BF>(acc, Xyzz<BF>::load(partial + (size_t)(t0 + k) * 128));
        uint8_t* const rec = buckets29 + (size_t)wide_true_bucket(key, low) * B29_BYTES;
        if (acc.is_identity()) store_b29_identity<BF>(rec); else store_b29<BF>(rec, xyzz_to29<BF>(acc));
    }
    if (nhot == 0) return;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, quad = lane >> 2;
    const u32 gw = blockIdx.x * (blockDim.x >> 6) + wave, GW = gridDim.x * (blockDim.x >> 6);
    u32 cbase = 0;                                         // first chunk slot of bucket h: the running sum of the chunk counts (the same in every wave)
    for (u32 h = 0; h < nhot; h++) {
        const u32 key = big[2 + h];
        const u32 t0 = toff[key], nt = toff[key + 1] - t0, nch = (nt + CHUNK - 1) / CHUNK;
        for (u32 c = (gw + GW - cbase % GW) % GW; c < nch; c += GW) {      // (chunk slot cbase + c belongs to wave (cbase + c) mod GW: thirteen buckets of 256 chunks spread over all waves, not over the first 256 thirteen times)
            const u32 lo = c * CHUNK, hi = lo + CHUNK < nt ? lo + CHUNK : nt;
            Fe<BF> acc = quad_identity<BF>();
            for (u32 k = lo + quad; k < hi; k += 16) acc = quad_add<BF>(acc, quad_load<BF>(partial + (size_t)(t0 + k) * 128));
            for (int d = 8; d >= 1; d >>= 1) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));
            if (lane < 4) quad_store<BF>(chunk_out + (size_t)(cbase + c) * 128, acc);
            __threadfence();                               // the chunk sum is visible device-wide before the arrival is counted
            u32 arrived = 0;
            if (lane == 0) arrived = __hip_atomic_fetch_add(&big[2 + cap + h], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            arrived = __shfl(arrived, 0, 64);
            if (arrived == nch) {                          // (wave-uniform) last one in: add up the chunk sums
                __threadfence();
                Fe<BF> o = quad_identity<BF>();
                for (u32 k = quad; k < nch; k += 16) o = quad_add<BF>(o, quad_load<BF>(chunk_out + (size_t)(cbase + k) * 128));
                for (int d = 8; d >= 1; d >>= 1) o = quad_add<BF>(o, quad_shfl_down<BF>(o, d));
                if (lane < 4) quad_store<BF>(total + 128 * wave, o);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    const Xyzz<BF> v = Xyzz<BF>::load(total + 128 * wave);
                    uint8_t* const rec = buckets29 + (size_t)wide_true_bucket(key, low) * B29_BYTES;
                    if (v.is_identity()) store_b29_identity<BF>(rec); else store_b29<BF>(rec, xyzz_to29<BF>(v));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        cbase += nch;
    }
}
// work item `out` of a group (the index of its output record): first bucket and bucket stride of its chunk
__device__ __forceinline__ void wide_a1_item(const WideGeom& g, u32 out, u32& t0, u32& tstride) {
    const u32 per = g.nb >> g.rlog;                        // items per plane
    if (out < per) { t0 = out << g.rlog; tstride = 1u; return; }
    const u32 cpm = (1u << g.hi) >> g.rlog, v = out - per, b = v / cpm, s = v - b * cpm;      // chunks per lo-digit marginal
    t0 = ((s << g.rlog) << g.lo) + b; tstride = 1u << g.lo;
}
// a record whose zz limb 0 is all ones (no normalised limb is): "this chunk's lazy sum met a possible exceptional case" -- k_wide_a2 redoes the chunk exactly
template <class F>
__device__ __forceinline__ bool is_marker29(const Acc29<F>& a) { return a.zz.v[0] == 0xffffffffu; }
template <class BF>
__global__ void __launch_bounds__(256)
k_wide_a1(const uint8_t* __restrict__ buckets29, WideGeom g, u32 ngroups, uint8_t* __restrict__ out29) {
    KH_HIGH_PRIO();
    const u32 per = g.nb >> g.rlog, items = 2u * per;
    const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= items * ngroups) return;
    const u32 q = gid / items, u = gid - q * items;
    // thread u -> output record: plane 0 in order; plane 1 with the lo digit fastest across lanes (neighbouring lanes read neighbouring buckets)
    u32 out = u;
    if (u >= per) { const u32 v = u - per, b = v & ((1u << g.lo) - 1u), sc = v >> g.lo; out = per + b * ((1u << g.hi) >> g.rlog) + sc; }
    u32 t0, ts;
    wide_a1_item(g, out, t0, ts);
    const uint8_t* B = buckets29 + ((size_t)q * g.nb + t0) * B29_BYTES;
    Acc29<BF> acc;
    bool have = false, ok = true;
    const u32 r = 1u << g.rlog;
#pragma unroll 1
    for (u32 i = 0; i < r; i++) {
        const Acc29<BF> b = load_b29<BF>(B + (size_t)i * ts * B29_BYTES);
        if (is_identity29<BF>(b)) continue;
        if (!have) { acc = b; have = true; continue; }
        ok = add29<BF>(acc, b);
        if (!ok) break;
    }
    uint8_t* o = out29 + ((size_t)q * items + out) * B29_BYTES;
    if (!ok) { acc.zz.v[0] = 0xffffffffu; store_b29<BF>(o, acc); return; }
    if (have) store_b29<BF>(o, acc); else store_b29_identity<BF>(o);
}
// the exact sum of one chunk (k_wide_a2's rare path; kept out of line so that its registers do not weigh on the kernel)
template <class BF>
__device__ __attribute__((noinline)) Xyzz<BF> wide_chunk_exact(const uint8_t* __restrict__ buckets29g, WideGeom g, u32 out) {
    u32 t0, ts;
    wide_a1_item(g, out, t0, ts);
    Xyzz<BF> acc = Xyzz<BF>::identity();
    const u32 r = 1u << g.rlog;
    for (u32 i = 0; i < r; i++) {
        const Acc29<BF> b = load_b29<BF>(buckets29g + ((size_t)t0 + (size_t)i * ts) * B29_BYTES);
        if (!is_identity29<BF>(b)) acc = add<BF>(acc, xyzz_from29<BF>(b));
    }
    return acc;
}
template <class BF>
__global__ void __launch_bounds__(64)
k_wide_a2(const uint8_t* __restrict__ in29, const uint8_t* __restrict__ buckets29, WideGeom g, uint8_t* __restrict__ outw) {
    KH_HIGH_PRIO();
    __shared__ u32 stage[32][64];                        // [word of the XYZZ record][lane]
    const u32 slot = blockIdx.x, q = blockIdx.y, plane = slot >> g.lo, sidx = slot & ((1u << g.lo) - 1u);
    const u32 per = g.nb >> g.rlog, items = 2u * per;
    const u32 lane = threadIdx.x, role = lane & 3u, quad = lane >> 2;
    uint8_t* o = outw + (((size_t)q * 2 + plane) * (1u << g.lo) + sidx) * 128;
    u32 cnt, first;
    if (plane == 0) {
        const u32 a = sidx + 1u;
        if (a >= (1u << g.hi)) { if (lane < 4) quad_store<BF>(o, Fe<BF>::zero()); return; }
        cnt = (1u << g.lo) >> g.rlog; first = a * cnt;
    } else { cnt = (1u << g.hi) >> g.rlog; first = per + sidx * cnt; }
    const uint8_t* src = in29 + ((size_t)q * items + first) * B29_BYTES;
    Xyzz<BF> a = Xyzz<BF>::identity();
    for (u32 k = lane; k < cnt; k += 64) {
        const Acc29<BF> b = load_b29<BF>(src + (size_t)k * B29_BYTES);
        if (__builtin_expect(is_marker29<BF>(b), 0)) a = add<BF>(a, wide_chunk_exact<BF>(buckets29 + (size_t)q * g.nb * B29_BYTES, g, first + k));
        else if (!is_identity29<BF>(b)) a = add<BF>(a, xyzz_from29<BF>(b));
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { stage[k][lane] = a.x.v[k]; stage[8 + k][lane] = a.y.v[k]; stage[16 + k][lane] = a.zz.v[k]; stage[24 + k][lane] = a.zzz.v[k]; }
    __syncthreads();
    auto parked = [&](u32 point) {
        Fe<BF> r;
#pragma unroll
        for (int k = 0; k < 8; k++) r.v[k] = stage[role * 8 + k][point];
        return r;
    };
    Fe<BF> acc = parked(quad);
    for (u32 k = 1; k < 4; k++) acc = quad_add<BF>(acc, parked(quad + 16 * k));
    for (int d = 8; d >= 1; d >>= 1) acc = quad_add<BF>(acc, quad_shfl_down<BF>(acc, d));
    if (lane < 4) quad_store<BF>(o, acc);
}

// ------------------------------------------------------------------------------------ precomputed window tables
// tables[w][i] = 2^(c*w) * P_i in affine form (w = 0 is the basis itself).  Thread per point:
// (W-1) x c doublings in XYZZ, then ONE inversion per point (Montgomery's trick over its W-1
// ZZZ values) to normalise.  `scratch` holds the (W-1) x n intermediate XYZZ points.
template <class BF>
__global__ void __launch_bounds__(256)
k_precompute(uint8_t* __restrict__ tables, const uint8_t* __restrict__ inf, size_t n, int c, int W, uint8_t* __restrict__ scratch) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (inf && inf[i]) {                   // multiples of the identity: never used (digits are zeroed), keep zeros
        for (int w = 1; w < W; w++) { Fe<BF> z = Fe<BF>::zero(); z.store(tables + ((size_t)w * n + i) * 64); z.store(tables + ((size_t)w * n + i) * 64 + 32); }
        return;
    }
    Xyzz<BF> P = Xyzz<BF>::from_affine(Aff<BF>::load(tables + i * 64));
    Fe<BF> prod = Fe<BF>::one();
    for (int w = 1; w < W; w++) {
        for (int k = 0; k < c; k++) P = dbl<BF>(P);
        P.store(scratch + ((size_t)(w - 1) * n + i) * 128);
        // running product of the ZZZ's, kept in the (now free) slot of the affine table: x-slot of table w
        prod = mul<BF>(prod, P.zzz);
        prod.store(tables + ((size_t)w * n + i) * 64);
    }
    Fe<BF> iv = inv<BF>(prod);             // 1 / (zzz_1 * ... * zzz_{W-1})
    for (int w = W - 1; w >= 1; w--) {
        Xyzz<BF> Q = Xyzz<BF>::load(scratch + ((size_t)(w - 1) * n + i) * 128);
        Fe<BF> izzz = iv;
        if (w > 1) izzz = mul<BF>(iv, Fe<BF>::load(tables + ((size_t)(w - 1) * n + i) * 64));   // times prefix product
        iv = mul<BF>(iv, Q.zzz);
        Fe<BF> izz = sqr<BF>(mul<BF>(izzz, Q.zz));       // (ZZ/ZZZ)^2 = 1/ZZ
        Fe<BF> x = mul<BF>(Q.x, izz), y = mul<BF>(Q.y, izzz);
        x.store(tables + ((size_t)w * n + i) * 64);
        y.store(tables + ((size_t)w * n + i) * 64 + 32);
    }
}

// ------------------------------------------------------------------------------------ host driver
struct VestaCfg { typedef FqParams Base; typedef FpParams Scalar; };
struct PallasCfg { typedef FpParams Base; typedef FqParams Scalar; };

static std::atomic<size_t>& wide_min_n_cell() {
    static std::atomic<size_t> v{getenv("KH_WIDE_MIN_N") ? (size_t)strtoull(getenv("KH_WIDE_MIN_N"), nullptr, 0) : ((size_t)1 << 19)};
    return v;
}
// the staged scatter of k_part2_sort: [0] entries per pass in LDS (KH_PART2_STAGE, 0 = the direct scatter), [1] the most passes a partition may take (KH_PART2_MAXPASS)
static std::atomic<u32>& sort_staging_cell(int i) {
    static std::atomic<u32> v[2] = {{getenv("KH_PART2_STAGE") ? (u32)atoi(getenv("KH_PART2_STAGE")) : 28672u}, {getenv("KH_PART2_MAXPASS") ? (u32)atoi(getenv("KH_PART2_MAXPASS")) : 2u}};
    return v[i];
}
void msm_set_sort_staging(unsigned entries, unsigned max_passes) { sort_staging_cell(0).store(entries); sort_staging_cell(1).store(max_passes); }
size_t msm_wide_min_n() { const size_t v = wide_min_n_cell().load(); return v ? v : ~(size_t)0; }
void msm_set_wide_min_n(size_t n) { wide_min_n_cell().store(n); }
int msm_pick_window(size_t n) {
    int lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = lg - 3;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
}

struct CaptureGuard {              // never leave a stream in capture mode on an error path
    hipStream_t s; bool active = false;
    ~CaptureGuard() { if (active) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(s, &g); if (g) (void)hipGraphDestroy(g); } }
};
// the task-length table of pick_K for a launch with `nkeys` buckets on `cus` compute units (cached per bucket count)
static KTab task_length_table(size_t nkeys, u32 cus) {
    static std::mutex mu;
    static std::map<std::pair<size_t, u32>, KTab> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({nkeys, cus});
    if (it != cache.end()) return it->second;
    KTab t{};
    for (int idx = 0; idx < 48; idx++) {
        const double m = pow(2.0, (idx - 8) / 4.0);
        std::vector<double> p(1, exp(-m));                                   // Poisson(m), up to the far tail
        for (int c = 1; c < (int)(m + 12 * sqrt(m) + 24); c++) p.push_back(p.back() * m / c);
        auto tasks_per_bucket = [&](int K) { double s = 0; for (size_t c = 1; c < p.size(); c++) s += p[c] * (double)((c + K - 1) / K); return s; };
        if (ceil(ceil((double)nkeys * tasks_per_bucket(8) / 256.0) / cus) > 6) { t.k[idx] = 0; continue; }     // a big job: the total decides
        double best = 1e300; int bk = 8;
        for (int K = 4; K <= 32; K++) {
            const double tpb = tasks_per_bucket(K);
            const double blocks = ceil((double)nkeys * tpb / 256.0);
            const double len = std::min<double>(K, m + 2.3 * sqrt(m));
            const double cost = ceil(blocks / cus) * len + 2.0 * tpb;
            if (cost < best - 1e-9) { best = cost; bk = K; }
        }
        t.k[idx] = (uint8_t)bk;
    }
    cache[{nkeys, cus}] = t;
    return t;
}
static inline uint64_t fnv(uint64_t h, uint64_t v) { for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xff; h *= 0x100000001b3ull; } return h; }

template <class CFG>
static int msm_enqueue_t(Context& Ctx, MsmSlot& C, const MsmBasis& basis, size_t offset, const u64* scalars_dev, size_t n, size_t k,
                         int mont, int curve, int use_graph, const MsmHostScalars* hs) {
    typedef typename CFG::Base BF; typedef typename CFG::Scalar SF;
    hipStream_t s = C.stream;
    // wide windows (a second table set, c = 20: msm.hpp) for big single MSMs: 13 instead of 16 additions per scalar.  Needs the wide sort's
    // record (<= 65,536 digits per pass-A block, <= 1024 blocks).
    const bool wide = basis.wide_pts && basis.wide_c > 16 && basis.wide_c - 9 <= (int)PART2_MAXLOW && n >= msm_wide_min_n() && k <= 4 &&
                      (size_t)((256 + basis.wide_c - 1) / basis.wide_c) * n <= ((size_t)1 << 26);
    // MSM_SPREAD_SCALARS: the caller vouches that the scalars are spread like random ones (the opening rounds: products with Fiat-Shamir challenges), so
    // no bucket collects a large share of the entries: the two-phase hot-bucket kernels are not launched (two empty dependent launches per MSM, ~20 us of
    // the ~340 us of an opening round) and a bucket's quad sums however many task partials it finds -- correct for any input, slow for a skewed one.
    // The promise is checked, not trusted: a bucket with more than SPREAD_MAX_NT task partials (a constant or sparse polynomial from a
    // degenerate or adversarial witness puts n / K of them into one bucket per window) is left empty and reported through a pinned word; msm_finish re-runs
    // the job with the hot-bucket kernels and suspends the hint for the rest of the opening (Context::spread_suspended, reset by kh_ipa_begin).
    static const bool spread_on = !(getenv("KH_NO_SPREAD_HINT") && atoi(getenv("KH_NO_SPREAD_HINT")) != 0);
    static const u32 SPREAD_MAX_NT = getenv("KH_SPREAD_MAX_NT") ? (u32)atoi(getenv("KH_SPREAD_MAX_NT")) : 256u;
    const bool spread = spread_on && (use_graph & MSM_SPREAD_SCALARS) != 0 && !Ctx.spread_suspended;
    const void* const tab_pts = wide ? basis.wide_pts : basis.pts;
    const int c = wide ? basis.wide_c : (basis.precomp_c ? basis.precomp_c : msm_pick_window(n));
    const bool glv = basis.glv && !wide && basis.precomp_c != 0;      // rows 0 .. W/2-1: 2^(c w) P, rows W/2 .. W-1: phi of them (csrc/rebase.hip)
    const int W = glv ? 2 * ((128 + c - 1) / c) : (256 + c - 1) / c;
    const u32 nb = 1u << (c - 1);
    const int precomp = (wide || basis.precomp_c) ? 1 : 0;
    // slices per (window, msm): enough blocks to fill the chip (~512), no more -- the per-slice histograms
    // cost nkeys x W x S words of traffic in k_key_totals, which dominates the sort of a batch
    static const size_t slice_div = getenv("KH_SLICE_DIV") ? (size_t)atol(getenv("KH_SLICE_DIV")) : 32768;
    int S = (int)(n / slice_div); if (S < 1) S = 1; if (S > 16) S = 16;
    { int want = (int)(512 / ((size_t)W * k)); if (want < 1) want = 1; if (S > want) S = want; }
    const size_t tab_stride = basis.stride ? basis.stride : basis.n;
    SortGeom g{n, nb, S, W, precomp, tab_stride, offset, basis.batch_stride};
    const size_t ngroups = precomp ? k : k * (size_t)W;
    const int Sq = precomp ? W * S : S;
    const size_t nkeys = ngroups * nb;
    const size_t M = n * (size_t)W * k;                 // upper bound on entries
    // Task size: the accumulation is one long dependent chain per thread, so a launch that needs
    // 1.1 "rounds" of resident threads pays a whole extra chain at 10 % occupancy (measured: 1.80 ms
    // instead of 1.72 ms at 2^20).  Size K so that all tasks are resident at once (4 waves/SIMD =
    // 1024 threads per CU) whenever the bucket count allows it.
    static const size_t room_waves = getenv("KH_ROOM_WAVES") ? (size_t)atoi(getenv("KH_ROOM_WAVES")) : 4;     // waves per SIMD the task count is sized for
    const size_t cap = (size_t)Ctx.num_cus * 256 * room_waves;
    size_t room_sz = cap / 4;                              // many buckets: about one task per bucket anyway
    if (nkeys < cap / 2) room_sz = cap - cap / 16 - nkeys / 2;   // ~ half of the buckets add a remainder task
    const u32 room = (u32)room_sz;                          // K = clamp(ceil(entries / room), 8, MAX_K), on the device
    static const u32 kmin = getenv("KH_KMIN") ? (u32)atoi(getenv("KH_KMIN")) : 8u;
    static const bool ktab_on = !getenv("KH_KMIN") && !(getenv("KH_KTAB") && atoi(getenv("KH_KTAB")) == 0);
    const KTab ktab = ktab_on && precomp ? task_length_table(nkeys, (u32)Ctx.num_cus) : KTab{};
    const size_t max_tasks = M / (kmin < 4 ? kmin : 4) + nkeys + 1;          // bound for the smallest K (the table's entries are >= 4)
    KH_REQUIRE(M < ((size_t)1 << 31) && (tab_stride * (size_t)(precomp ? W : 1) + basis.batch_stride * k) < ((size_t)1 << 31), "MSM too large for 31-bit entry indices (n=%zu k=%zu)", n, k);

    int rc;
    if ((rc = C.ws_digits.reserve(M * sizeof(int32_t)))) return rc;
    if ((rc = C.ws_hist.reserve(nkeys * Sq * sizeof(u32)))) return rc;
    if ((rc = C.ws_cnt.reserve((nkeys + 2) * sizeof(u32)))) return rc;
    if ((rc = C.ws_off.reserve((nkeys + 2) * sizeof(u32)))) return rc;
    if ((rc = C.ws_ntask.reserve((nkeys + 2) * sizeof(u32)))) return rc;
    if ((rc = C.ws_toff.reserve((nkeys + 2) * sizeof(u32)))) return rc;
    if ((rc = C.ws_entries.reserve((M + 1) * sizeof(u32)))) return rc;
    if ((rc = C.ws_partial.reserve(max_tasks * 128))) return rc;
    if ((rc = C.ws_handed.reserve((max_tasks + 2) * sizeof(u32)))) return rc;
    if ((rc = C.ws_buckets.reserve(nkeys * 128))) return rc;
    const size_t bigcap = nkeys + max_tasks / CHUNK + 2;      // big buckets <= nkeys; chunk items <= tasks/512 + nkeys
    if ((rc = C.ws_biglist.reserve((2 + 4 * bigcap) * sizeof(u32)))) return rc;
    if ((rc = C.ws_chunks.reserve(bigcap * 128))) return rc;
    if ((rc = C.ws_order.reserve((2 * (MAX_K + 1) + 3 * (nkeys + 2)) * sizeof(u32)))) return rc;
    WideGeom wg{};
    // wide-path knobs (defaults = measured best, tools/wide_sweep.py): ranks of one partition side by side in the accumulation order; blocks of the
    // accumulation per CU (held down with dynamic LDS); log2 of the chunk length of the first reduction level
    static const u32 wide_og = getenv("KH_WIDE_OG") ? (u32)std::max(1, atoi(getenv("KH_WIDE_OG"))) : 16u;
    static const u32 wide_acc_blocks = getenv("KH_WIDE_ACC_BLOCKS") ? (u32)std::max(1, atoi(getenv("KH_WIDE_ACC_BLOCKS"))) : 3u;
    static const u32 wide_rlog = getenv("KH_WIDE_RLOG") ? (u32)atoi(getenv("KH_WIDE_RLOG")) : 4u;
    // dynamic LDS that leaves room for exactly wide_acc_blocks blocks on a CU, from the device's own LDS size (160 KB on gfx950), never more than one block
    // may ask for: where the request cannot hold the count down the kernel simply runs at its register-limited occupancy
    size_t wide_acc_lds = wide_acc_blocks >= 4 ? 0 : (Ctx.lds_per_cu / (wide_acc_blocks + 1) + 1024) & ~(size_t)1023;
    if (wide_acc_lds > Ctx.lds_per_block) wide_acc_lds = Ctx.lds_per_block & ~(size_t)1023;
    // the narrow accumulation held to fewer blocks per CU the same way (KH_ACC_BLOCKS, experiment: room for the latency kernels of OTHER provers' chains; 0 = no limit)
    static const unsigned acc_blocks = getenv("KH_ACC_BLOCKS") ? (unsigned)atoi(getenv("KH_ACC_BLOCKS")) : 0u;
    size_t acc_lds = (acc_blocks == 0 || acc_blocks >= 4) ? 0 : (Ctx.lds_per_cu / (acc_blocks + 1) + 1024) & ~(size_t)1023;
    if (acc_lds > 65536) acc_lds = 65536;
    if (wide) {
        // k_part2_sort interleaves `og` consecutive ranks of a partition: og must divide the 2^low buckets of a partition; k_wide_a1 / _a2 cut both digit
        // planes into chunks of 2^rlog buckets: rlog <= min(lo, hi)
        KH_REQUIRE(wide_og >= 1 && (wide_og & (wide_og - 1)) == 0 && wide_og <= (1u << ((u32)c - 9)), "KH_WIDE_OG = %u must be a power of two <= %u", wide_og, 1u << ((u32)c - 9));
        KH_REQUIRE(wide_rlog >= 1 && wide_rlog <= std::min((u32)c / 2, (u32)(c - 1) - (u32)c / 2), "KH_WIDE_RLOG = %u must be in 1..%u", wide_rlog, std::min((u32)c / 2, (u32)(c - 1) - (u32)c / 2));                                            // bucket = (hi digit, lo digit); chunks of 2^rlog buckets in the first reduction level
        if ((rc = C.ws_xlist.reserve((3 * max_tasks + 8) * sizeof(u32)))) return rc;       // (key, chunk) pairs of the extra chunks, then the split buckets' keys
        if ((rc = C.ws_handed.reserve((2 * max_tasks + 4) * sizeof(u32)))) return rc;
        wg.nb = nb; wg.lo = (u32)c / 2; wg.hi = (u32)(c - 1) - wg.lo; wg.rlog = wide_rlog;
        if ((rc = C.ws_b29.reserve(nkeys * B29_BYTES))) return rc;
        if ((rc = C.ws_a1.reserve(ngroups * 2 * (size_t)(nb >> wg.rlog) * B29_BYTES))) return rc;
        if ((rc = C.ws_a2.reserve(ngroups * 2 * ((size_t)1 << wg.lo) * 128))) return rc;
    }
    // segment length of the weighted reduction: one bucket per thread (a 15-bit double-and-add each)
    // is the shortest chain, but its work grows with the bucket count -- for batches use running
    // sums over m buckets (2 additions per bucket + one double-and-add per segment), keeping about
    // one wave per SIMD busy
    u32 m = 1;
    while (m < 16 && (ngroups * (size_t)nb) / m > 65536) m <<= 1;
    if (!precomp && m < 4) m = 4;
    if (nb < m) m = nb;
    const u32 nseg = nb / m;
    const u32 nblk1 = (nseg + SUM_BLK - 1) / SUM_BLK;
    // one to four MSMs over the tables: digit marginals instead (k_marginals), three fields of <= 5 bits
    static const size_t marg_max = getenv("KH_MARG_MAX") ? (size_t)atol(getenv("KH_MARG_MAX")) : 4096;
    const u32 planes = wide ? 3u : ((precomp && ngroups <= marg_max && c >= 7 && c <= 16 && !getenv("KH_NO_PLANES")) ? 3u : 0u);
    MargGeom mg{};
    if (planes) {                                          // (wide: the tail runs over the two marginal groups of 2^lo slots each, k_wide_a2)
        const u32 bits = wide ? wg.lo : (u32)c - 1, f0 = (bits + 2) / 3, f1 = (bits - f0 + 1) / 2, f2 = bits - f0 - f1;
        mg.nb = wide ? 1u << wg.lo : nb; mg.sh[0] = 0; mg.wd[0] = f0; mg.sh[1] = f0; mg.wd[1] = f1; mg.sh[2] = f0 + f1; mg.wd[2] = f2;
    }
    const size_t tail_groups = wide ? 2 * ngroups : ngroups;
    const size_t nout = planes ? tail_groups * planes : ngroups;
    if ((rc = C.ws_seg.reserve(std::max(ngroups * (size_t)(nseg + nblk1), nout * (size_t)32) * 128))) return rc;
    if ((rc = C.ws_out.reserve(nout * 128))) return rc;
    if (C.pinned_cap < nout * 128) {                     // host staging of the group sums; the host part runs in msm_finish
        if (C.pinned) (void)hipHostFree(C.pinned);
        C.pinned = nullptr; C.pinned_cap = 0;
        // coherent + mapped, explicitly: the completion-by-flag path reads these lines with no runtime synchronisation in between
        C.pinned_coherent = hipHostMalloc(&C.pinned, nout * 128 + 4096, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
        if (!C.pinned_coherent) { (void)hipGetLastError(); C.pinned = nullptr; KH_HIP(hipHostMalloc(&C.pinned, nout * 128 + 4096, hipHostMallocDefault)); }
        C.pinned_cap = nout * 128 + 4096;
    }
    if (Ctx.once("msm_attr")) {
        KH_HIP(hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        KH_HIP(hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        KH_HIP(hipFuncSetAttribute((const void*)k_sort_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    }
    // small jobs over the tables: the whole sort in one launch (k_sort_fused)
    static const bool fused_off = getenv("KH_NO_FUSED_SORT") != nullptr;
    static const size_t fused_max = getenv("KH_FUSED_MAX") ? (size_t)atol(getenv("KH_FUSED_MAX")) : ((size_t)1 << 22);
    // bounded spin of its grid barriers: 20 ms of the 100 MHz wall clock by default (a barrier normally waits < 50 us)
    static const unsigned long long fused_spin_ticks = 100ull * (getenv("KH_FUSED_SPIN_US") ? (unsigned long long)atoll(getenv("KH_FUSED_SPIN_US")) : 20000ull);
    FusedGeom fg{};
    // (co-residency of the spinning blocks is what makes the grid barriers safe: 64 blocks x 4 jobs in flight need 128 CUs -- not in a partitioned mode)
    bool fused = precomp && !wide && !fused_off && !Ctx.fused_disabled && k <= 4 && M < fused_max && nb >= 2048 && Ctx.num_cus >= 128;
    if (fused) {
        fg.n = (u32)n; fg.nb = nb; fg.W = (u32)W; fg.k = (u32)k; fg.bpg = (u32)std::min<size_t>(FUSED_G, FUSED_B / (2 * k)); fg.nkeys = (u32)nkeys;
        fg.split = 2; fg.sub = nb / fg.split;              // 64 blocks of 64 KB LDS: two per CU, so four (even eight) jobs in flight stay co-resident
        fg.room = room; fg.kmin = kmin; fg.ktab = ktab; fg.pt_stride = tab_stride; fg.pt_offset = offset; fg.pt_batch = basis.batch_stride;
        const size_t threads = (size_t)fg.bpg * k * fg.split * FUSED_T;
        fg.kpt = (u32)((nkeys + 1 + threads - 1) / threads);
        const size_t tot4 = (size_t)W * n / 4, chunk4 = (tot4 + fg.bpg - 1) / fg.bpg + 1;
        if (fg.kpt > (u32)FUSED_KPT || n < 4 || ((size_t)W * n) % 4 || chunk4 > (size_t)FUSED_V * FUSED_T || fg.sub > (u32)FUSED_C * FUSED_T) fused = false;
    }
    // large jobs over the tables: the two-pass partitioned sort (k_part_*)
    static const bool part_off = getenv("KH_NO_PART_SORT") != nullptr;
    const bool part = wide || (precomp && !fused && !part_off && nb == ((u32)PART_P << PART_LOW) && n <= ((size_t)1 << 20) && W <= 16 && M >= ((size_t)1 << 21));
    const u32 part_low = wide ? (u32)c - 9 : (u32)PART_LOW;
    const u32 part_nblk = wide ? (u32)std::max<size_t>(std::max<size_t>(1, std::min<size_t>(256, 512 / k)), ((size_t)W * n + 65535) / 65536)
                               : (part ? (u32)std::max<size_t>(1, std::min<size_t>(256, 512 / k)) : 0);
    const size_t part_size = (size_t)k * PART_P * part_nblk;
    if (part) {
        if ((rc = C.ws_hist.reserve((part_size + 1) * sizeof(u32)))) return rc;
        if ((rc = C.ws_cnt.reserve((part_size + 1) * sizeof(u32)))) return rc;
        if ((rc = C.ws_mid.reserve(M * sizeof(u32)))) return rc;
    }
    if (fused) {
        if ((rc = C.ws_hist.reserve((size_t)k * fg.bpg * nb * sizeof(u32)))) return rc;
        if (!C.ws_sync.p) {
            if ((rc = C.ws_sync.reserve((2 + 2 * FUSED_B + 32 + 2) * sizeof(u32)))) return rc;
            KH_HIP(hipMemsetAsync(C.ws_sync.p, 0, (2 + 2 * FUSED_B + 32 + 2) * sizeof(u32), s));
        }
        if (!C.host_abort) {
            void* hp = nullptr;
            KH_HIP(hipHostMalloc(&hp, 64, hipHostMallocCoherent | hipHostMallocMapped));
            C.host_abort = (volatile uint32_t*)hp; *C.host_abort = 0;
        }
    }
    if (spread && !C.spread_abort) {
        void* hp = nullptr;
        KH_HIP(hipHostMalloc(&hp, 64, hipHostMallocCoherent | hipHostMallocMapped));
        C.spread_abort = (volatile uint32_t*)hp; *C.spread_abort = 0;
    }
    // completion by flag (MsmSlot::done_flag): pinned word + two device words, once per slot
    static const bool flag_off = getenv("KH_NO_DONE_FLAG") != nullptr;
    bool flag_on = !flag_off && (wide || planes) && !C.flag_unavailable && C.pinned_coherent;
    if (flag_on && !C.done_flag) {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {      // no coherent word: completion by event on this slot
            (void)hipGetLastError(); C.flag_unavailable = true; flag_on = false;
        } else {
            C.done_flag = (volatile uint32_t*)hp; *C.done_flag = 0; C.done_expect = 0;
            if ((rc = C.ws_done.reserve(64))) return rc;
            KH_HIP(hipMemsetAsync(C.ws_done.p, 0, 64, s));           // ordered in front of the slot's first launch that counts in it
        }
    }
    // hipGraph replay / capture (opt-in by the caller; the key covers everything the launches bake in)
    static const bool graphs_off = getenv("KH_NO_GRAPH") != nullptr;
    uint64_t key = 0;
    CaptureGuard gcap{s};
    const bool timers_were = C.timer.enabled;
    if ((use_graph & MSM_REPEATS) && !graphs_off && s != Ctx.stream) {       // never capture on the main stream: other host threads synchronise and launch on it
        key = 0xcbf29ce484222325ull;
        const uint64_t parts[] = {(uint64_t)(uintptr_t)basis.pts, (uint64_t)(uintptr_t)basis.inf, basis.n, basis.stride, basis.batch_stride, (uint64_t)basis.precomp_c,
                                  offset, (uint64_t)(uintptr_t)scalars_dev, n, k, (uint64_t)mont, (uint64_t)curve, (uint64_t)(uintptr_t)C.pinned,
                                  (uint64_t)(uintptr_t)C.ws_digits.p, (uint64_t)(uintptr_t)C.ws_hist.p, (uint64_t)(uintptr_t)C.ws_cnt.p, (uint64_t)(uintptr_t)C.ws_off.p,
                                  (uint64_t)(uintptr_t)C.ws_ntask.p, (uint64_t)(uintptr_t)C.ws_toff.p, (uint64_t)(uintptr_t)C.ws_entries.p, (uint64_t)(uintptr_t)C.ws_partial.p,
                                  (uint64_t)(uintptr_t)C.ws_buckets.p, (uint64_t)(uintptr_t)C.ws_seg.p, (uint64_t)(uintptr_t)C.ws_out.p, (uint64_t)(uintptr_t)C.ws_scan_tmp.p,
                                  (uint64_t)(uintptr_t)C.ws_biglist.p, (uint64_t)(uintptr_t)C.ws_order.p, (uint64_t)(uintptr_t)C.ws_chunks.p,
                                  (uint64_t)(uintptr_t)C.ws_handed.p, (uint64_t)(uintptr_t)C.ws_sync.p, (uint64_t)fused, (uint64_t)(uintptr_t)C.ws_mid.p, (uint64_t)part,
                                  (uint64_t)(uintptr_t)tab_pts, (uint64_t)wide, (uint64_t)spread, (uint64_t)(uintptr_t)C.ws_xlist.p, (uint64_t)(uintptr_t)C.ws_b29.p, (uint64_t)(uintptr_t)C.ws_a1.p, (uint64_t)(uintptr_t)C.ws_a2.p, (uint64_t)(uintptr_t)C.ws_done.p, (uint64_t)flag_on, (uint64_t)(use_graph & MSM_LATENCY), (uint64_t)glv,
                                  ((uint64_t)sort_staging_cell(0).load() << 8) | sort_staging_cell(1).load(), DevBuf::generation().load()};
        for (uint64_t v : parts) key = fnv(key, v);
        if (C.gexec && C.gkey == key) {                    // replay
            C.fused_used = C.g_fused;
            C.retry = {basis.pts, basis.inf, basis.n, basis.stride, basis.batch_stride, basis.precomp_c, offset, scalars_dev, n, k, mont, curve, basis.glv};
            // (the host's launch count moves BEFORE the launch that will move the device's: whatever fails in between, the host is never
            // behind -- a stale equality would end a later wait early -- and a host that is ahead only falls back to the event, then resyncs)
            C.spread_used = C.g_spread;
            C.done_by_flag = C.g_done_by_flag; if (C.done_by_flag) C.done_expect++;
            KH_HIP(hipGraphLaunch(C.gexec, s));
            counter(CNT_GRAPH_REPLAY)++;
            KH_HIP(hipEventRecord(C.done, s));
            C.busy = true; C.owner = std::this_thread::get_id(); C.ticket = Ctx.next_ticket++;
            C.curve = curve; C.W = C.g_W; C.c = C.g_c; C.precomp = C.g_precomp; C.k = k; C.ngroups = C.g_ngroups; C.planes = C.g_planes;
            C.plane_shift[0] = C.g_shift[0]; C.plane_shift[1] = C.g_shift[1]; C.wide_lo = C.g_wide_lo;
            C.timer.n = 0;                                 // no per-phase events inside a graph
            return KH_OK;
        }
        if (C.gseen == key) {                              // second time: every workspace is sized, capture this one
            if (C.gexec) { (void)hipGraphExecDestroy(C.gexec); C.gexec = nullptr; }
            C.timer.enabled = false;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) gcap.active = true;
            else { (void)hipGetLastError(); C.timer.enabled = timers_were; }
        } else {
            C.gseen = key;
        }
    }

    C.timer.begin(s);
    // 1 digits
    if (hs && hs->nev > 0 && k == 1 && !gcap.active) {    // upload and digit pass chunk by chunk (MsmHostScalars)
        for (int ch = 0; ch < hs->nev; ch++) {
            const size_t c0 = n * (size_t)ch / hs->nev, c1 = n * (size_t)(ch + 1) / hs->nev;
            if (c1 == c0) continue;
            KH_HIP(hipMemcpyAsync((char*)scalars_dev + c0 * 32, (const char*)hs->host + c0 * 32, (c1 - c0) * 32, hipMemcpyHostToDevice, hs->cs));
            KH_HIP(hipEventRecord(hs->ev[ch], hs->cs));
            KH_HIP(hipStreamWaitEvent(s, hs->ev[ch], 0));
            hipLaunchKernelGGL((k_digits<SF>), dim3((unsigned)((c1 - c0 + 255) / 256), 1u), dim3(256), 0, s,
                               scalars_dev, basis.inf, offset, basis.batch_stride, n, mont, c, W, C.ws_digits.as<int32_t>(), c0, c1);
        }
    } else if (glv)
    hipLaunchKernelGGL((k_digits_glv<SF>), dim3((unsigned)((n + 255) / 256), (unsigned)k), dim3(256), 0, s,
                       scalars_dev, basis.inf, offset, basis.batch_stride, n, mont, c, W, C.ws_digits.as<int32_t>());
    else
    hipLaunchKernelGGL((k_digits<SF>), dim3((unsigned)((n + 255) / 256), (unsigned)k), dim3(256), 0, s,
                       scalars_dev, basis.inf, offset, basis.batch_stride, n, mont, c, W, C.ws_digits.as<int32_t>(), (size_t)0, n);
    C.timer.mark("digits", s);
    u32* len_hist = C.ws_order.as<u32>();                       // [MAX_K+1] histogram, [MAX_K+1] cursors, then order / rnt / roff
    u32* cursor = len_hist + (MAX_K + 1);
    u32* order = cursor + (MAX_K + 1);
    u32* rnt = order + (nkeys + 2);
    u32* roff = rnt + (nkeys + 2);
    if (fused) {
        order = nullptr; roff = C.ws_toff.as<u32>();
        u32* sy = C.ws_sync.as<u32>();
        hipLaunchKernelGGL(k_sort_fused, dim3(fg.bpg * fg.k * fg.split), dim3(FUSED_T), (size_t)fg.sub * sizeof(u32), s, C.ws_digits.as<int32_t>(), fg,
                           C.ws_hist.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), C.ws_entries.as<u32>(), C.ws_handed.as<u32>(),
                           C.ws_biglist.as<u32>(), sy, sy + 2, C.host_abort, fused_spin_ticks);
        C.timer.mark("sort", s);
    } else if (part) {
        const u32 tot_e = (u32)((size_t)W * n);
        const dim3 pgrid(part_nblk, (unsigned)k);
        hipLaunchKernelGGL(k_part_hist, pgrid, dim3(PART_T), 0, s, C.ws_digits.as<int32_t>(), tot_e, part_nblk, wide ? 0u : part_low, C.ws_hist.as<u32>());
        C.timer.mark("histogram", s);
        if ((rc = exclusive_scan_u32(C.ws_hist.as<u32>(), C.ws_cnt.as<u32>(), part_size + 1, C.ws_scan_tmp, s))) return rc;
        C.timer.mark("scan", s);
        if (wide) {
            hipLaunchKernelGGL(k_part2_scatter, pgrid, dim3(PART_T), 0, s, C.ws_digits.as<int32_t>(), tot_e, part_nblk, part_low, C.ws_cnt.as<u32>(), C.ws_mid.as<u32>(), C.ws_xlist.as<u32>(), C.ws_biglist.as<u32>());
            // ... and the task plan (toff, the length-ranked order, roff): partition-local in pass B, made global by k_wide_fixup
            const WideTasks wt{room, kmin, (u32)nkeys, ktab};
            u32* const slist = C.ws_xlist.as<u32>() + 2 + 2 * max_tasks;
            u32* const ptot = C.ws_ntask.as<u32>();        // k * 256 partition task totals (the narrow path's per-key task counts: unused here)
            // staged scatter (round 6): KH_PART2_STAGE = entries per pass in LDS (0: the direct scatter); the dynamic LDS also holds the planning phase's lists
            const u32 part2_stage = sort_staging_cell(0).load(), part2_maxpass = std::min<u32>(PART2_MAXPASS, sort_staging_cell(1).load());
            // (clamped to the 128 KB the attribute below asks for; an MSM whose partitions cannot make it in max_pass passes does not ask for the LDS at all)
            u32 stage_now = std::min<u32>(part2_stage, (131072u - (1u << PART2_MAXLOW)) / 4u);
            if (stage_now <= 2048u || (size_t)W * n / PART_P > (size_t)part2_maxpass * (stage_now - 1024u)) stage_now = 0;
            const u32 stage_words = std::max<u32>(stage_now, 3u << PART2_MAXLOW);
            const size_t part2_lds = (size_t)stage_words * 4 + (stage_now ? ((size_t)1 << PART2_MAXLOW) : 0);
            if (Ctx.once("part2_attr")) KH_HIP(hipFuncSetAttribute((const void*)k_part2_sort, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            hipLaunchKernelGGL(k_part2_sort, dim3(PART_P, (unsigned)k), dim3(PART_T), part2_lds, s, C.ws_mid.as<u32>(), C.ws_cnt.as<u32>(), part_nblk, part_low, nb, tot_e, (u32)n,
                               tab_stride, offset, basis.batch_stride, (u32)(k - 1), part_size, wt, C.ws_off.as<u32>(), C.ws_entries.as<u32>(),
                               C.ws_toff.as<u32>(), order, ptot, C.ws_xlist.as<u32>(), slist, C.ws_b29.as<uint8_t>(), wide_og, stage_now, part2_maxpass, C.ws_biglist.as<u32>(), (u32)bigcap, WIDE_HOT_NT);
            hipLaunchKernelGGL(k_wide_fixup, dim3((unsigned)(nkeys / 256 + 1)), dim3(256), 0, s, C.ws_toff.as<u32>(), ptot, (u32)(k * PART_P), part_low, (u32)nkeys,
                               C.ws_handed.as<u32>(), C.ws_biglist.as<u32>());
        } else {
        hipLaunchKernelGGL(k_part_scatter, pgrid, dim3(PART_T), 0, s, C.ws_digits.as<int32_t>(), tot_e, (u32)n, part_nblk, C.ws_cnt.as<u32>(), C.ws_mid.as<u32>());
        hipLaunchKernelGGL(k_part_sort, dim3(PART_P, (unsigned)k), dim3(PART_T), 0, s, C.ws_mid.as<u32>(), C.ws_cnt.as<u32>(), part_nblk, nb,
                           tab_stride, offset, basis.batch_stride, (u32)(k - 1), C.ws_off.as<u32>(), C.ws_entries.as<u32>());
        }
        C.timer.mark("scatter", s);
    } else {
        // 2 histogram
        size_t lds = (size_t)nb * sizeof(u32);
        dim3 sgrid((unsigned)S, (unsigned)W, (unsigned)k);
        hipLaunchKernelGGL(k_hist, sgrid, dim3(1024), lds, s, C.ws_digits.as<int32_t>(), g, C.ws_hist.as<u32>());
        hipLaunchKernelGGL(k_key_totals, dim3((unsigned)((nkeys + 1 + 255) / 256)), dim3(256), 0, s,
                           C.ws_hist.as<u32>(), nb, Sq, nkeys, C.ws_cnt.as<u32>());
        C.timer.mark("histogram", s);
        // 3 scan
        if ((rc = exclusive_scan_u32(C.ws_cnt.as<u32>(), C.ws_off.as<u32>(), nkeys + 1, C.ws_scan_tmp, s))) return rc;
        C.timer.mark("scan", s);
        // 4 scatter
        hipLaunchKernelGGL(k_scatter, sgrid, dim3(1024), lds, s, C.ws_digits.as<int32_t>(), g, C.ws_hist.as<u32>(),
                           C.ws_off.as<u32>(), C.ws_entries.as<u32>());
        C.timer.mark("scatter", s);
    }
    if (!fused && !wide) {
        // tasks
        hipLaunchKernelGGL(k_task_reset, dim3(1), dim3(320), 0, s, len_hist, C.ws_biglist.as<u32>());      // (two memset nodes cost the host ~10 us each to queue)
        // (1024-thread blocks: the per-(block, length) global atomics of k_ntask / k_len_rank serialise on ~40 hot addresses -- 25 us each at 256 threads per block)
        hipLaunchKernelGGL(k_ntask, dim3((unsigned)((nkeys + 1 + 1023) / 1024)), dim3(1024), 0, s, C.ws_off.as<u32>(), nkeys, room, kmin, ktab, C.ws_ntask.as<u32>(), len_hist, C.ws_handed.as<u32>());
        if ((rc = exclusive_scan_u32(C.ws_ntask.as<u32>(), C.ws_toff.as<u32>(), nkeys + 1, C.ws_scan_tmp, s))) return rc;
        static const int rank_min_log = getenv("KH_RANK_MIN_LOG") ? atoi(getenv("KH_RANK_MIN_LOG")) : 22;   // below ~4M entries the three extra launches cost more than the ordering saves
        if (M >= ((size_t)1 << rank_min_log)) {
            hipLaunchKernelGGL(k_len_starts, dim3(1), dim3(320), 0, s, len_hist, cursor);
            hipLaunchKernelGGL(k_len_rank, dim3((unsigned)((nkeys + 1 + 1023) / 1024)), dim3(1024), 0, s, C.ws_off.as<u32>(), C.ws_ntask.as<u32>(), nkeys, cursor, order, rnt);
            if ((rc = exclusive_scan_u32(rnt, roff, nkeys + 1, C.ws_scan_tmp, s))) return rc;
        } else {                     // small problems are launch-latency bound: keep the key order
            order = nullptr; roff = C.ws_toff.as<u32>();
        }
        C.timer.mark("tasks", s);
    }
    // 5 accumulate: the lazy 29-bit-limb kernel, then the exact kernel over the (almost always zero) tasks it handed over
    static const bool acc29 = !(getenv("KH_ACC29") && atoi(getenv("KH_ACC29")) == 0);
    const u32* abort_dev = fused ? C.ws_sync.as<u32>() + 2 + 2 * FUSED_B + 32 : nullptr;     // the fused sort's give-up word (sort_gave_up)
    const u32* handed = acc29 ? C.ws_handed.as<u32>() : nullptr;
    const dim3 agrid((unsigned)((max_tasks + 255) / 256));
    const u32 acc_prio = (use_graph & MSM_LATENCY) ? 1u : 0u;
    uint8_t* const b29 = wide ? C.ws_b29.as<uint8_t>() : nullptr;
    if (wide) {                                            // thread per bucket in pass B's interleaved length order (+ the listed extra chunks), then the exact redos
        const dim3 wgrid((unsigned)(nkeys / 256));
        auto kern = k_acc_wide29<BF>;
        if (C.timer.enabled && C.timer.created && !gcap.active) {
            hipExtLaunchKernelGGL(kern, wgrid, dim3(256), wide_acc_lds, s, C.timer.k0, C.timer.k1, 0, C.ws_entries.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), order, (u32)nkeys,
                                  (const uint8_t*)tab_pts, C.ws_partial.as<uint8_t>(), b29, part_low, C.ws_handed.as<u32>());
            C.timer.kname = "k_acc_wide29";
        } else
        hipLaunchKernelGGL(kern, wgrid, dim3(256), wide_acc_lds, s, C.ws_entries.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), order, (u32)nkeys,
                           (const uint8_t*)tab_pts, C.ws_partial.as<uint8_t>(), b29, part_low, C.ws_handed.as<u32>());
        // (the extra chunks: none for 2^19..2^21 uniform scalars; at 2^22 the 2^14 top-window buckets hold ~360 entries against K = 256 and split in two --
        //  16 K extra chunks, for which 32 blocks took 2 ms)
        const unsigned xgrid = M / nkeys >= 64 ? 2048u : WIDE_XB;
        hipLaunchKernelGGL((k_acc_wide_rest<BF>), dim3(xgrid), dim3(256), 0, s, C.ws_entries.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), C.ws_xlist.as<u32>(),
                           C.ws_handed.as<u32>(), (const uint8_t*)tab_pts, C.ws_partial.as<uint8_t>(), b29, part_low);
    } else
    if (acc29) {
        auto kern = k_accumulate29<BF>;
        if (C.timer.enabled && C.timer.created && !gcap.active) {     // the dominant kernel's own start / stop timestamps (bench.py roofline)
            hipExtLaunchKernelGGL(kern, agrid, dim3(256), acc_lds, s, C.timer.k0, C.timer.k1, 0,
                                  C.ws_entries.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), roff, order, nkeys,
                                  (const uint8_t*)tab_pts, C.ws_partial.as<uint8_t>(), C.ws_handed.as<u32>(), abort_dev, acc_prio);
            C.timer.kname = "k_accumulate29";
        } else
        hipLaunchKernelGGL(kern, agrid, dim3(256), acc_lds, s,
                           C.ws_entries.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), roff, order, nkeys,
                           (const uint8_t*)tab_pts, C.ws_partial.as<uint8_t>(), C.ws_handed.as<u32>(), abort_dev, acc_prio);
    } else if (C.timer.enabled && C.timer.created && !gcap.active) {
        hipExtLaunchKernelGGL((k_accumulate<BF>), agrid, dim3(256), 0, s, C.timer.k0, C.timer.k1, 0,
                              C.ws_entries.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), roff, order, nkeys,
                              (const uint8_t*)tab_pts, C.ws_partial.as<uint8_t>(), handed, abort_dev);
        C.timer.kname = "k_accumulate";
    } else
    hipLaunchKernelGGL((k_accumulate<BF>), agrid, dim3(256), 0, s,
                       C.ws_entries.as<u32>(), C.ws_off.as<u32>(), C.ws_toff.as<u32>(), roff, order, nkeys,
                       (const uint8_t*)tab_pts, C.ws_partial.as<uint8_t>(), handed, abort_dev);
    C.timer.mark("accumulate", s);
    // 6 bucket sums
    static const bool bsum_quad = !getenv("KH_NO_BSUM_QUAD");
    static const size_t bsum_maxg = getenv("KH_QUAD_MAXG") ? (size_t)atol(getenv("KH_QUAD_MAXG")) : 8;     // 5 / 7 / 8 MSMs of 2^16: 0.94 / 1.09 / 1.14 -> 0.85 / 1.05 / 1.09 ms against 4
    if (wide)                                              // (the split buckets' sums, hot ones included: one launch; 2^22: 16 K split buckets of two partials)
        hipLaunchKernelGGL((k_bucket_sum_wide<BF>), dim3(256), dim3(256), 0, s, C.ws_toff.as<u32>(), C.ws_xlist.as<u32>(), C.ws_xlist.as<u32>() + 2 + 2 * max_tasks,
                           C.ws_partial.as<uint8_t>(), b29, part_low, WIDE_HOT_NT, C.ws_biglist.as<u32>(), (u32)bigcap, C.ws_chunks.as<uint8_t>());
    else if (precomp && ngroups <= bsum_maxg && bsum_quad)
        hipLaunchKernelGGL((k_bucket_sum_q<BF>), dim3((unsigned)((4 * nkeys + 255) / 256)), dim3(256), 0, s,
                           C.ws_toff.as<u32>(), nkeys, C.ws_partial.as<uint8_t>(), C.ws_buckets.as<uint8_t>(), C.ws_biglist.as<u32>(),
                           bigcap, spread ? SPREAD_MAX_NT : 16u, abort_dev, spread ? (u32*)C.spread_abort : nullptr);
    else
    hipLaunchKernelGGL((k_bucket_sum<BF>), dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, s,
                       C.ws_toff.as<u32>(), nkeys, C.ws_partial.as<uint8_t>(), C.ws_buckets.as<uint8_t>(), C.ws_biglist.as<u32>(),
                       bigcap, nkeys <= 16384 ? 4u : 16u, abort_dev);
    if (!wide && !(spread && precomp && ngroups <= bsum_maxg && bsum_quad)) {     // (the quad kernel took every bucket: nothing was listed; the wide path's kernel sums its hot buckets itself)
    hipLaunchKernelGGL((k_bucket_chunk<BF>), dim3(2048), dim3(64), 0, s,
                       C.ws_toff.as<u32>(), C.ws_partial.as<uint8_t>(), C.ws_biglist.as<u32>(), bigcap, C.ws_chunks.as<uint8_t>(), abort_dev);
    hipLaunchKernelGGL((k_bucket_big<BF>), dim3(1024), dim3(256), 0, s,
                       C.ws_toff.as<u32>(), C.ws_chunks.as<uint8_t>(), C.ws_buckets.as<uint8_t>(), C.ws_biglist.as<u32>(), bigcap, abort_dev);
    }
    C.timer.mark("bucket_sum", s);
    // 7 reduce
    u32* const done_ws = flag_on ? C.ws_done.as<u32>() : nullptr; u32* const done_flag = flag_on ? (u32*)C.done_flag : nullptr;
    if (flag_on) C.done_expect++;                        // before the launch that stores it (see the replay branch); every flag_on job ends in k_marginal_fin_q
    bool direct_out = false;                              // latency path: the last kernel writes the (768-byte) result to host memory itself -- no copy node
    if (wide) {
        const u32 items = 2u * (nb >> wg.rlog) * (u32)ngroups;
        hipLaunchKernelGGL((k_wide_a1<BF>), dim3((items + 255) / 256), dim3(256), 0, s, b29, wg, (u32)ngroups, C.ws_a1.as<uint8_t>());
        C.timer.mark("reduce_a1", s);
        hipLaunchKernelGGL((k_wide_a2<BF>), dim3(2u << wg.lo, (unsigned)ngroups), dim3(64), 0, s, C.ws_a1.as<uint8_t>(), b29, wg, C.ws_a2.as<uint8_t>());
        hipLaunchKernelGGL((k_marginals_q<BF>), dim3(32, 3, (unsigned)tail_groups), dim3(256), 0, s, C.ws_a2.as<uint8_t>(), mg, C.ws_seg.as<uint8_t>());
        hipLaunchKernelGGL((k_marginal_fin_q<BF>), dim3(3, (unsigned)tail_groups), dim3(128), 0, s, C.ws_seg.as<uint8_t>(), mg, (uint8_t*)C.pinned, done_ws, done_flag);
        direct_out = true;
    } else if (planes) {
        // few groups: 256 threads per marginal (4 sequential additions + the tree: shortest chain); batches: one wave
        // per marginal (16 + 6 additions deep, but 2.2x less issue work -- batches are throughput-bound)
        static const int quad_threads = getenv("KH_QUAD") ? atoi(getenv("KH_QUAD")) : 256;   // 0: scalar additions
        static const size_t quad_maxg = getenv("KH_QUAD_MAXG") ? (size_t)atol(getenv("KH_QUAD_MAXG")) : 8;
        static const bool marg_hybrid = !(getenv("KH_MARG_HYBRID") && atoi(getenv("KH_MARG_HYBRID")) == 0) && !getenv("KH_QUAD");
        if (ngroups <= quad_maxg && quad_threads > 0) {    // latency path: lane-cooperative additions (coop.cuh)
            if (marg_hybrid && (mg.nb >> mg.wd[0]) >= 256)  // ... for the tree; one lane per addition where every lane has its own buckets (k_marginals_h)
                hipLaunchKernelGGL((k_marginals_h<BF>), dim3(32, 3, (unsigned)ngroups), dim3(256), 0, s, C.ws_buckets.as<uint8_t>(), mg, C.ws_seg.as<uint8_t>());
            else
            hipLaunchKernelGGL((k_marginals_q<BF>), dim3(32, 3, (unsigned)ngroups), dim3(quad_threads), 0, s, C.ws_buckets.as<uint8_t>(), mg, C.ws_seg.as<uint8_t>());
            hipLaunchKernelGGL((k_marginal_fin_q<BF>), dim3(3, (unsigned)ngroups), dim3(128), 0, s, C.ws_seg.as<uint8_t>(), mg, (uint8_t*)C.pinned, done_ws, done_flag);   // straight into the pinned host staging
            direct_out = true;
        } else {
        static const bool marg_h_batch = !(getenv("KH_MARG_H_BATCH") && atoi(getenv("KH_MARG_H_BATCH")) == 0);
        if (marg_h_batch && marg_hybrid && (mg.nb >> mg.wd[0]) >= 256)      // batches too: the same sums, the 6-level lane tree replaced by the quads' 4 + 4 steps
            hipLaunchKernelGGL((k_marginals_h<BF>), dim3(32, 3, (unsigned)ngroups), dim3(256), 0, s, C.ws_buckets.as<uint8_t>(), mg, C.ws_seg.as<uint8_t>());
        else
        hipLaunchKernelGGL((k_marginals<BF>), dim3(32, 3, (unsigned)ngroups), dim3(ngroups <= 4 ? 256 : 64), 0, s, C.ws_buckets.as<uint8_t>(), mg, C.ws_seg.as<uint8_t>());
        // the weighted tail is 3 x ngroups blocks however many MSMs there are: a pure latency chain (10 additions deep), so it takes the
        // lane-cooperative kernel for batches too (15 witness columns: 96 -> 35 us); the records have the same 128-byte layout
        static const bool fin_quad = !getenv("KH_NO_FIN_QUAD");
        if (fin_quad) { hipLaunchKernelGGL((k_marginal_fin_q<BF>), dim3(3, (unsigned)ngroups), dim3(128), 0, s, C.ws_seg.as<uint8_t>(), mg, (uint8_t*)C.pinned, done_ws, done_flag); direct_out = true; }
        else
        hipLaunchKernelGGL((k_marginal_fin<BF>), dim3(3, (unsigned)ngroups), dim3(64), 0, s, C.ws_seg.as<uint8_t>(), mg, C.ws_out.as<uint8_t>());
        }
    } else {
    hipLaunchKernelGGL((k_reduce_seg<BF>), dim3((unsigned)((ngroups * nseg + 127) / 128)), dim3(128), 0, s,
                       C.ws_buckets.as<uint8_t>(), nb, m, ngroups, C.ws_seg.as<uint8_t>());
    {
        uint8_t* lvl1 = C.ws_seg.as<uint8_t>() + ngroups * (size_t)nseg * 128;
        hipLaunchKernelGGL((k_sum_level<BF>), dim3(nblk1, (unsigned)ngroups), dim3(256), 0, s, C.ws_seg.as<uint8_t>(), nseg, nblk1, lvl1);
        hipLaunchKernelGGL((k_sum_final<BF>), dim3((unsigned)ngroups), dim3(nblk1 > 64 ? 256 : 64), 0, s, lvl1, nblk1, C.ws_out.as<uint8_t>());
    }
    }
    C.timer.mark("reduce", s);
    KH_HIP(hipGetLastError());
    // group sums -> pinned host staging
    if (!direct_out) KH_HIP(hipMemcpyAsync(C.pinned, C.ws_out.p, nout * 128, hipMemcpyDeviceToHost, s));
    if (gcap.active) {
        hipGraph_t g = nullptr;
        gcap.active = false;
        C.timer.enabled = timers_were; C.timer.n = 0;
        hipError_t e = hipStreamEndCapture(s, &g);
        if (e == hipSuccess && g) e = hipGraphInstantiate(&C.gexec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (e != hipSuccess) { C.gexec = nullptr; set_error("hipGraph capture of the MSM launch sequence failed: %s", hipGetErrorString(e)); return KH_E_DEVICE; }
        C.gkey = key; C.gnout = nout; C.gscalars = scalars_dev; C.g_fused = fused; C.g_spread = spread && !wide && precomp && ngroups <= bsum_maxg && bsum_quad;
        C.g_W = W; C.g_c = c; C.g_precomp = precomp; C.g_planes = (int)planes; C.g_shift[0] = (int)mg.wd[0]; C.g_shift[1] = (int)mg.wd[1]; C.g_ngroups = ngroups; C.g_wide_lo = wide ? (int)wg.lo : 0; C.g_done_by_flag = direct_out && flag_on;
        KH_HIP(hipGraphLaunch(C.gexec, s));
    }
    KH_HIP(hipEventRecord(C.done, s));
    C.done_by_flag = direct_out && flag_on;
    C.busy = true; C.owner = std::this_thread::get_id(); C.ticket = Ctx.next_ticket++;
    C.curve = curve; C.W = W; C.c = c; C.precomp = precomp; C.k = k; C.ngroups = ngroups; C.planes = (int)planes; C.plane_shift[0] = (int)mg.wd[0]; C.plane_shift[1] = (int)mg.wd[1]; C.wide_lo = wide ? (int)wg.lo : 0;
    C.fused_used = fused;
    C.spread_used = spread && !wide && precomp && ngroups <= bsum_maxg && bsum_quad;
    C.retry = {basis.pts, basis.inf, basis.n, basis.stride, basis.batch_stride, basis.precomp_c, offset, scalars_dev, n, k, mont, curve, basis.glv};
    return KH_OK;
}

int msm_enqueue(Context& C, MsmSlot& S, int curve, const MsmBasis& basis, size_t offset, const uint64_t* scalars_dev, size_t n, size_t k, int mont,
                int use_graph, const MsmHostScalars* hs) {
    if (n == 0 || k == 0) {          // nothing to launch: finish() emits k identities
        S.busy = true; S.owner = std::this_thread::get_id(); S.ticket = C.next_ticket++; S.curve = curve; S.k = k; S.ngroups = 0; S.W = 0; S.c = 0; S.precomp = 1; S.planes = 0; S.done_by_flag = false; S.fused_used = false; S.spread_used = false;
        KH_HIP(hipEventRecord(S.done, S.stream));
        return KH_OK;
    }
    if (hs) use_graph &= ~MSM_REPEATS;                     // (a captured launch sequence cannot contain the host's staging)
    if (curve == KH_CURVE_VESTA) return msm_enqueue_t<VestaCfg>(C, S, basis, offset, scalars_dev, n, k, mont, curve, use_graph, hs);
    return msm_enqueue_t<PallasCfg>(C, S, basis, offset, scalars_dev, n, k, mont, curve, use_graph, hs);
}

// 8 finish on the host: wait for the slot, Horner over the window sums (plain path), XYZZ -> affine
int msm_finish(Context& C, MsmSlot& S, uint64_t* out_xy, uint8_t* out_inf, bool flag_seen) {
    // (the per-phase events of a job -- tools only -- are read below: they need the event wait)
    if (!(flag_seen && S.done_by_flag && S.timer.n == 0)) KH_HIP(hipEventSynchronize(S.done));
    if (S.fused_used && S.host_abort && *S.host_abort) {
        // the one-launch sort gave up at a grid barrier (its blocks were not all resident: the GPU is shared with another process):
        // everything behind it ran on garbage.  Re-run this job with the multi-launch sort and stay on it.
        *S.host_abort = 0;
        if (!C.fused_disabled) fprintf(stderr, "libkimchi_hip: k_sort_fused could not get its blocks co-resident (shared GPU?): using the multi-launch sort from now on\n");
        C.fused_disabled = true;
        counter(CNT_FUSED_RETRY)++;
        KH_HIP(hipMemsetAsync(S.ws_sync.p, 0, (2 + 2 * FUSED_B + 32 + 2) * sizeof(u32), S.stream));
        const uint64_t ticket = S.ticket; const auto owner = S.owner;
        MsmBasis b; b.pts = S.retry.pts; b.inf = S.retry.inf; b.n = S.retry.bn; b.stride = S.retry.stride; b.batch_stride = S.retry.batch_stride; b.precomp_c = S.retry.precomp_c; b.glv = S.retry.glv;
        int rc = msm_enqueue(C, S, S.retry.curve, b, S.retry.offset, S.retry.scalars, S.retry.n, S.retry.k, S.retry.mont, 0);
        S.ticket = ticket; S.owner = owner; C.next_ticket--;
        if (rc) { S.busy = false; return rc; }
        KH_HIP(hipEventSynchronize(S.done));
    }
    if (S.spread_used && S.spread_abort && *S.spread_abort) {
        // MSM_SPREAD_SCALARS did not hold (a bucket with more task partials than a quad may sum in sequence): the buckets that said so were left empty.
        // Re-run this job with the hot-bucket kernels; the hint stays off until the caller begins its next opening.
        *S.spread_abort = 0;
        C.spread_suspended = true;
        counter(CNT_SPREAD_RETRY)++;
        const uint64_t ticket = S.ticket; const auto owner = S.owner;
        MsmBasis b; b.pts = S.retry.pts; b.inf = S.retry.inf; b.n = S.retry.bn; b.stride = S.retry.stride; b.batch_stride = S.retry.batch_stride; b.precomp_c = S.retry.precomp_c; b.glv = S.retry.glv;
        int rc = msm_enqueue(C, S, S.retry.curve, b, S.retry.offset, S.retry.scalars, S.retry.n, S.retry.k, S.retry.mont, 0);
        S.ticket = ticket; S.owner = owner; C.next_ticket--;
        if (rc) { S.busy = false; return rc; }
        KH_HIP(hipEventSynchronize(S.done));
    }
    S.busy = false;
    static const bool fused_dbg = getenv("KH_FUSED_DEBUG") != nullptr;
    if (fused_dbg && S.ws_sync.p) {                       // phase timestamps of the last k_sort_fused on this slot (block 0)
        unsigned long long ts[12]; (void)hipMemcpy(ts, S.ws_sync.as<u32>() + 2 + 2 * FUSED_B, sizeof(ts), hipMemcpyDeviceToHost);
        fprintf(stderr, "k_sort_fused phases (us):"); for (int i = 1; i < 9; i++) fprintf(stderr, " %.1f", (double)(ts[i] - ts[i - 1]) / 100.0);
        fprintf(stderr, " | p3: %.1f %.1f %.1f %.1f\n", (double)(ts[9] - ts[4]) / 100.0, (double)(ts[10] - ts[9]) / 100.0, (double)(ts[11] - ts[10]) / 100.0, (double)(ts[5] - ts[11]) / 100.0);
    }
    collect_timings(C, S.timer);
    if (S.ngroups == 0) {
        for (size_t j = 0; j < S.k; j++) { memset(out_xy + 8 * j, 0, 64); out_inf[j] = 1; }
        return KH_OK;
    }
    const khost::xyzz* res = (const khost::xyzz*)S.pinned;
    const MsmSlot* Sp = &S;
    static thread_local std::vector<khost::xyzz> totals;
    totals.resize(S.k);
    khost::xyzz* tot = totals.data();
    auto finish_one = [res, Sp, tot](size_t j) {
        khost::Crv crv(Sp->curve);
        khost::xyzz& total = tot[j];
        if (Sp->wide_lo) {                            // two marginal groups per MSM (k_wide_a2): 2^lo * fold(H) + fold(L)
            for (int g2 = 0; g2 < 2; g2++) {
                const khost::xyzz* r3 = res + (j * 2 + g2) * 3;
                khost::xyzz f = r3[2];
                for (int t = 0; t < Sp->plane_shift[1]; t++) f = crv.dbl(f);
                f = crv.add(f, r3[1]);
                for (int t = 0; t < Sp->plane_shift[0]; t++) f = crv.dbl(f);
                f = crv.add(f, r3[0]);
                if (g2 == 0) { for (int t = 0; t < Sp->wide_lo; t++) f = crv.dbl(f); total = f; }
                else total = crv.add(total, f);
            }
        } else if (Sp->precomp && Sp->planes) {       // (M2 2^f1 + M1) 2^f0 + M0 over the three digit marginals
            total = res[j * 3 + 2];
            for (int t = 0; t < Sp->plane_shift[1]; t++) total = crv.dbl(total);
            total = crv.add(total, res[j * 3 + 1]);
            for (int t = 0; t < Sp->plane_shift[0]; t++) total = crv.dbl(total);
            total = crv.add(total, res[j * 3]);
        } else if (Sp->precomp) total = res[j];
        else {                                        // Horner over the window sums: ~256 doublings (~70 us)
            total = crv.identity();
            for (int w = Sp->W - 1; w >= 0; w--) {
                for (int t = 0; t < Sp->c; t++) total = crv.dbl(total);
                total = crv.add(total, res[j * Sp->W + w]);
            }
        }
    };
    if ((!S.precomp && S.k >= 2) || (S.planes && S.k >= 16)) {      // split the Horner folds with the helper thread (a hand-over costs ~10 us: not for a few doublings)
        const size_t half = S.k / 2, kk = S.k;
        HostHelper& hh = host_helper(&C);
        hh.run([finish_one, half, kk] { for (size_t j = half; j < kk; j++) finish_one(j); });
        for (size_t j = 0; j < half; j++) finish_one(j);
        hh.wait();
    } else {
        for (size_t j = 0; j < S.k; j++) finish_one(j);
    }
    // XYZZ -> affine for all k results with ONE field inversion (Montgomery's trick over the ZZZ's): an inversion is
    // ~13 us on the host, 0.3 ms for a batch of 23 commitments
    khost::Crv crv(S.curve); const khost::Fld& F = crv.F;
    static thread_local std::vector<khost::fe> pre;
    pre.resize(S.k + 1);
    pre[0] = F.f.one;
    for (size_t j = 0; j < S.k; j++) pre[j + 1] = crv.is_identity(tot[j]) ? pre[j] : F.mul(pre[j], tot[j].zzz);
    khost::fe inv = F.inv(pre[S.k]);
    for (size_t j = S.k; j-- > 0;) {
        if (crv.is_identity(tot[j])) { memset(out_xy + 8 * j, 0, 64); out_inf[j] = 1; continue; }
        const khost::fe izzz = F.mul(inv, pre[j]);
        inv = F.mul(inv, tot[j].zzz);
        const khost::fe izz = F.sqr(F.mul(izzz, tot[j].zz));          // (ZZ / ZZZ)^2 = 1 / ZZ
        const khost::fe x = F.mul(tot[j].x, izz), y = F.mul(tot[j].y, izzz);
        memcpy(out_xy + 8 * j, &x, 32); memcpy(out_xy + 8 * j + 4, &y, 32);
        out_inf[j] = 0;
    }
    return KH_OK;
}

// the same on a caller's stream, with a caller's scratch buffer ((W - 1) x n x 128 bytes), no synchronisation: the opening's rebased tables (rebase.hip)
int msm_precompute_on(hipStream_t s, int curve, void* tables, size_t n, int c, void* scratch) {
    const int W = (256 + c - 1) / c;
    dim3 grid((unsigned)((n + 63) / 64));                  // small blocks: the chain of 255 doublings per point is pure latency, one wave per SIMD is the fastest
    if (curve == KH_CURVE_VESTA) hipLaunchKernelGGL((k_precompute<FqParams>), grid, dim3(64), 0, s, (uint8_t*)tables, (const uint8_t*)nullptr, n, c, W, (uint8_t*)scratch);
    else hipLaunchKernelGGL((k_precompute<FpParams>), grid, dim3(64), 0, s, (uint8_t*)tables, (const uint8_t*)nullptr, n, c, W, (uint8_t*)scratch);
    KH_HIP(hipGetLastError());
    return KH_OK;
}
int msm_precompute(Context& C, int curve, void* tables, const uint8_t* inf, size_t n, int c) {
    const int W = (256 + c - 1) / c;
    DevBuf& scratch = C.scratch("precompute");
    int rc = scratch.reserve((size_t)(W - 1) * n * 128); if (rc) return rc;
    dim3 grid((unsigned)((n + 255) / 256));
    if (curve == KH_CURVE_VESTA) hipLaunchKernelGGL((k_precompute<FqParams>), grid, dim3(256), 0, C.stream, (uint8_t*)tables, inf, n, c, W, scratch.as<uint8_t>());
    else hipLaunchKernelGGL((k_precompute<FpParams>), grid, dim3(256), 0, C.stream, (uint8_t*)tables, inf, n, c, W, scratch.as<uint8_t>());
    KH_HIP(hipGetLastError());
    KH_HIP(hipStreamSynchronize(C.stream));
    if (scratch.cap > ((size_t)1 << 30)) scratch.release();     // do not pin GBs of scratch
    return KH_OK;
}

// ------------------------------------------------------------------------------------ debug hooks
template <class F>
__global__ void k_debug_field(int op, const u64* a, const u64* b, u64* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> x = Fe<F>::load(a + 4 * i), y = Fe<F>::zero(), r;
    if (b) y = Fe<F>::load(b + 4 * i);
    switch (op) {
        case 0: r = mul<F>(x, y); break;
        case 1: r = add<F>(x, y); break;
        case 2: r = sub<F>(x, y); break;
        case 3: r = to_mont<F>(x); break;
        case 4: r = from_mont<F>(x); break;
        case 5: r = sqr<F>(x); break;
        case 7: r = from29<F>(mul29<F>(to29<F>(x), to29<F>(y))); break;        // the 29-bit-limb arithmetic of field29.cuh
        case 8: r = from29<F>(sqr29<F>(to29<F>(x))); break;
        case 9: r = unpack29<F>(pack29<F, 0>(x)); break;
        case 10: r = from29<F>(mul29<F>(pack29<F, 5>(x), to29<F>(y))); break;     // the unreduced 32 X form of a table point
        default: r = neg<F>(x); break;
    }
    r.store(out + 4 * i);
}
int debug_field_op(Context& C, int field, int op, const u64* a, const u64* b, u64* out, size_t n) {
    dim3 grid((unsigned)((n + 255) / 256));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_debug_field<FpParams>), grid, dim3(256), 0, C.stream, op, a, b, out, n);
    else hipLaunchKernelGGL((k_debug_field<FqParams>), grid, dim3(256), 0, C.stream, op, a, b, out, n);
    KH_HIP(hipGetLastError());
    return KH_OK;
}
// out = XYZZ (128 B) so the host can normalise
template <class F>
__global__ void k_debug_point(int op, const u64* p, const uint8_t* pinf, const u64* q, const uint8_t* qinf, uint8_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<F> P = Aff<F>::load(p + 8 * i), Q = Aff<F>::load(q + 8 * i);
    Xyzz<F> a = (pinf && pinf[i]) ? Xyzz<F>::identity() : Xyzz<F>::from_affine(P);
    Xyzz<F> b = (qinf && qinf[i]) ? Xyzz<F>::identity() : Xyzz<F>::from_affine(Q);
    Xyzz<F> r;
    if (op == 0) r = add<F>(a, b);
    else if (op == 1) r = dbl<F>(a);
    else if (op == 2) r = (qinf && qinf[i]) ? a : madd<F>(a, Q, false);
    else if (op == 3) { Xyzz<F> d2 = dbl<F>(a); r = (qinf && qinf[i]) ? d2 : madd<F>(d2, Q, true); }   // 2P - Q: non-trivial ZZ into madd
    else {        // op 6: (2P + Q) through madd29 (non-trivial ZZ); a record of 0xff bytes = "handed to the exact path"
        Xyzz<F> d2 = dbl<F>(a);
        Acc29<F> A; A.x = to29<F>(d2.x); A.y = to29<F>(d2.y); A.zz = to29<F>(d2.zz); A.zzz = to29<F>(d2.zzz);
        const bool ok = madd29<F>(A, pack29<F, 5>(Q.x), pack29<F, 5>(Q.y));
        r.x = from29<F>(A.x); r.y = from29<F>(A.y); r.zz = from29<F>(A.zz); r.zzz = from29<F>(A.zzz);
        if (!ok) { Fe<F> ff; _Pragma("unroll") for (int k = 0; k < 8; k++) ff.v[k] = 0xffffffffu; r.x = ff; r.y = ff; r.zz = ff; r.zzz = ff; }
    }
    r.store(out + 128 * i);
}
// ops 4, 5: the lane-cooperative addition (coop.cuh), four threads per pair; op 5 doubles both operands first so that the
// inputs have non-trivial ZZ / ZZZ
template <class F>
__global__ void k_debug_quad(int op, const u64* p, const uint8_t* pinf, const u64* q, const uint8_t* qinf, uint8_t* out, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i = t >> 2; const int role = (int)(t & 3);
    const bool live = i < n;
    if (!live) i = n - 1;                                  // keep whole quads / waves in the shuffles
    Aff<F> P = Aff<F>::load(p + 8 * i), Q = Aff<F>::load(q + 8 * i);
    Xyzz<F> a = (pinf && pinf[i]) ? Xyzz<F>::identity() : Xyzz<F>::from_affine(P);
    Xyzz<F> b = (qinf && qinf[i]) ? Xyzz<F>::identity() : Xyzz<F>::from_affine(Q);
    if (op == 5) { a = dbl<F>(a); b = dbl<F>(b); }
    const Fe<F> ac = role == 0 ? a.x : (role == 1 ? a.y : (role == 2 ? a.zz : a.zzz));
    const Fe<F> bc = role == 0 ? b.x : (role == 1 ? b.y : (role == 2 ? b.zz : b.zzz));
    const Fe<F> r = quad_add<F>(ac, bc);
    if (live) quad_store<F>(out + 128 * i, r);
}
int debug_point_op(Context& C, int curve, int op, const u64* p, const uint8_t* pinf, const u64* q, const uint8_t* qinf, uint8_t* out, size_t n) {
    if (op == 4 || op == 5) {
        dim3 qgrid((unsigned)((4 * n + 255) / 256));
        if (curve == KH_CURVE_VESTA) hipLaunchKernelGGL((k_debug_quad<FqParams>), qgrid, dim3(256), 0, C.stream, op, p, pinf, q, qinf, out, n);
        else hipLaunchKernelGGL((k_debug_quad<FpParams>), qgrid, dim3(256), 0, C.stream, op, p, pinf, q, qinf, out, n);
        KH_HIP(hipGetLastError());
        return KH_OK;
    }
    dim3 grid((unsigned)((n + 255) / 256));
    if (curve == KH_CURVE_VESTA) hipLaunchKernelGGL((k_debug_point<FqParams>), grid, dim3(256), 0, C.stream, op, p, pinf, q, qinf, out, n);
    else hipLaunchKernelGGL((k_debug_point<FpParams>), grid, dim3(256), 0, C.stream, op, p, pinf, q, qinf, out, n);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

}  // namespace kh
