// ntt.hip -- radix-2 NTT / iNTT / low-degree extension over the Pasta fields on gfx950.
//
// Replaces ark-poly's Radix2EvaluationDomain::{fft_in_place, ifft_in_place} reached from
// Evaluations::interpolate (kimchi/src/prover.rs:289,377,907,1163, permutation.rs:571,
// poly-commitment/src/utils.rs:195-196) and DensePolynomial::evaluate_over_domain_by_ref(d8)
// (kimchi/src/circuits/constraints.rs:490-495).  Semantics: natural order in and
// out, omega_N = (5^T)^(2^(32-k)), forward out[j] = sum_i a_i w^(ij), inverse includes 1/N.
// A DFT is a unique result, so only outputs are compared with the reference; the schedule
// is a multi-pass decimation-in-frequency decomposition built for the MI355X:
//
//   N = R_1 * R_2 * ... * R_P  (each R <= 256).  Pass p runs R_p-point sub-transforms on
//   tiles of T columns (R*T = 1024 elements, four per thread).  A thread keeps its four elements
//   in registers and does two radix-2 stages per step; between steps the tile goes through LDS
//   (32 KiB, limb-plane SoA so every access is a conflict-free ds_read/write_b64); the first
//   step loads straight from global memory and the last stores straight to it, multiplying by
//   the inter-pass twiddle w_L^(b*k) on the way out.  Stage twiddles are read from LDS.
//   Global accesses are T*32-byte contiguous runs.  Positions after the passes are
//   digit-reversed; the LAST pass undoes that while writing (its tiles take T rows with
//   consecutive top digit so the natural-order stores are T*32-byte runs too).
//   An inverse transform of two or more passes carries its 1/N in the first pass's inter-pass
//   twiddles (a second, scaled table), so the scaling costs no product of its own.
//   The low-degree extension n -> n*2^b is the same machinery with a virtual first digit
//   (the coset index r): the first pass reads the n coefficients and multiplies by
//   w_(n 2^b)^(i*r) while loading, so the zero-padded 7/8 of the input never exists in HBM.
#include <map>
#include <tuple>

#include "common.hpp"
#include "field.cuh"
#include "host_ec.hpp"
#include "msm.hpp"

namespace kh {

// 2^10 elements (32 KiB of LDS) and 256 threads per workgroup: four workgroups per CU.  The butterflies are VALU-bound, so a pass
// lasts as long as the busiest CU's share, ceil(tiles / CUs) tiles; against 2^11-element tiles the finer grain wins where the tile
// count is a small multiple of the CU count (19 columns of 2^16: 0.131 -> 0.116 ms) and ties elsewhere (2^22: 0.512 vs 0.509 ms).
static constexpr int NTT_THREADS = 256;
static constexpr int NTT_LOG_TILE = 10;                 // R*T elements per workgroup = 4 per thread
// Largest sub-transform of a pass, 2^9 (KH_NTT_MAX_LOGR, 4..10).  Round 5 measured the two-pass splits the kernel had never been given: 2^18 = 512 x 512
// instead of 64^3 is 9 % faster (0.0503 -> 0.0457 ms); 2^19 = 1024 x 512 (one column per tile in the 1024-point pass: 32-byte accesses at a 16 KB
// stride) does NOT pay -- iNTT 2^19 0.0792 -> 0.0781 ms, the 16-column extension 2^16 -> 2^19 0.731 -> 0.7405 ms against the 0.72 the experiment was
// to beat: the inter-pass twiddle product it saves is cancelled by the lost coalescing -- so 2^19 stays 128 x 64 x 64 (gpurun_out / profiles: r05_ntt_split.txt).
static constexpr int NTT_MAX_LOGR = 9;

struct PassArgs {
    const u64* src; u64* dst; const u64* tw; const u64* tw_out;
    u32 log_ntot, log_n, log_blow, log_L, log_R, log_T;
    u32 first, last, src_is_coeffs, scale, scale_out;
    u32 nd; u32 dig[6];          // last pass: log-sizes of the row-index digits, most significant first
    u32 inv_n[8];                // N^-1 (Montgomery), used when scale or scale_out is set
    u64 batch;
};

template <class F>
__device__ __forceinline__ Fe<F> lds_get(const u64* pl, u32 stride, u32 idx) {
    Fe<F> r;
#pragma unroll
    for (int k = 0; k < 4; k++) { u64 w = pl[k * stride + idx]; r.v[2 * k] = (u32)w; r.v[2 * k + 1] = (u32)(w >> 32); }
    return r;
}
template <class F>
__device__ __forceinline__ void lds_put(u64* pl, u32 stride, u32 idx, const Fe<F>& a) {
#pragma unroll
    for (int k = 0; k < 4; k++) pl[k * stride + idx] = (u64)a.v[2 * k] | ((u64)a.v[2 * k + 1] << 32);
}
// w^e from the half table (N/2 entries): w^(e) = -w^(e - N/2) for e >= N/2
template <class F>
__device__ __forceinline__ Fe<F> tw_get(const u64* tw, u32 log_ntot, u64 e) {
    u64 half = (u64)1 << (log_ntot - 1);
    e &= ((u64)1 << log_ntot) - 1;
    bool ng = e >= half;
    if (ng) e -= half;
    Fe<F> w = Fe<F>::load(tw + 4 * e);
    return ng ? neg<F>(w) : w;
}
__device__ __forceinline__ u32 bitrev(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// One workgroup transforms a tile of T columns x R points.  Every thread owns FOUR elements per step and keeps them in
// registers: a step is two radix-2 stages (a radix-4 butterfly on the quad {lo, lo+q, lo+2q, lo+3q}), or a single stage on two
// pairs when the stage count is odd (taken first).  The first step reads its quad straight from global memory and the last one
// stores straight to it, so a pass of 8 stages makes 3 trips through LDS and 3 barriers instead of 9 and 9.
template <class F, int THREADS>
__global__ void __launch_bounds__(THREADS)
k_ntt_pass(PassArgs A) {
    extern __shared__ u64 lds[];
    const u32 R = 1u << A.log_R, T = 1u << A.log_T, RT = R * T;
    u64* data = lds;                       // 4 planes x RT
    u64* twl = lds + 4 * RT;               // 4 planes x (R/2) stage twiddles
    const u32 tid = threadIdx.x;
    const u64 ntot = (u64)1 << A.log_ntot, n = (u64)1 << A.log_n;
    const u32 log_B = A.log_L - A.log_R;
    const u64 B = (u64)1 << log_B;

    // ---- decode the tile
    u64 tile = blockIdx.x;
    u64 batch_idx = 0, k0 = 0, sa = 0, bt = 0, rest = 0, dt = 0, Mrows = 1;
    const bool row_mode = A.last != 0;
    const bool flat_rows = row_mode && A.nd == 0;          // single pass, no virtual digit: tiles run across the batch
    if (!row_mode) {
        u64 tiles_per_item = ntot >> (A.log_R + A.log_T);
        batch_idx = tile / tiles_per_item; u64 r = tile % tiles_per_item;
        u64 bt_count = B >> A.log_T;
        bt = r % bt_count; r /= bt_count;
        u64 subs = n >> A.log_L;
        sa = r % subs; k0 = r / subs;
    } else if (!flat_rows) {
        u64 rows = ntot >> A.log_R;
        u64 tiles_per_item = rows >> A.log_T;
        batch_idx = tile / tiles_per_item; u64 r = tile % tiles_per_item;
        Mrows = rows >> A.dig[0];
        rest = r % Mrows; dt = r / Mrows;
    }
    const u64 total_rows_flat = A.batch;             // flat mode: one row per batch item

    // ---- stage twiddles w_R^i = w_Ntot^(i * Ntot/R), i < R/2
    for (u32 i = tid; i < R / 2; i += THREADS) {
        Fe<F> w = Fe<F>::load(A.tw + 4 * ((u64)i << (A.log_ntot - A.log_R)));
        lds_put<F>(twl, R / 2, i, w);
    }

    // ---- the four elements of this thread (LDS index n1*T + t) for the first step
    const u32 nq = RT >= 4 ? RT / 4 : 1;
    const bool active = tid < nq;
    int lh = (int)A.log_R - 1;                       // upper stage of the coming step
    const bool odd = (A.log_R & 1) != 0;
    u32 li[4];
    bool ok01 = active, ok23 = active;
    if (odd) {                                       // two pairs of stage lh: (lo, lo+h)
        const u32 h = 1u << lh;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            u32 pi = tid + p * nq;
            u32 t = pi & (T - 1), pr = pi >> A.log_T;
            u32 i = pr & (h - 1), grp = pr >> lh;
            u32 lo = (grp << (lh + 1)) + i;
            li[2 * p] = lo * T + t; li[2 * p + 1] = (lo + h) * T + t;
        }
        ok01 = active && tid < RT / 2;
        ok23 = active && tid + nq < RT / 2;
    } else {
        const u32 q = 1u << (lh - 1);
        u32 t = tid & (T - 1), pr = tid >> A.log_T;
        u32 i = pr & (q - 1), grp = pr >> (lh - 1);
        u32 lo = (grp << (lh + 1)) + i;
#pragma unroll
        for (int j = 0; j < 4; j++) li[j] = (lo + j * q) * T + t;
    }

    // ---- load them (all four loads issued before anything waits on one)
    Fe<F> x[4];
    u64 ex[4];                                        // exponent of the extension's coset twiddle (0 = none)
    {
        u64 spos[4]; bool have[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 n1 = li[j] >> A.log_T, t = li[j] & (T - 1);
            have[j] = j < 2 ? ok01 : ok23; ex[j] = 0;
            if (!row_mode) {
                u64 inner = sa * ((u64)1 << A.log_L) + (u64)n1 * B + bt * T + t;     // position inside the inner NTT
                spos[j] = A.src_is_coeffs ? (batch_idx * n + inner) : (batch_idx * ntot + k0 * n + inner);
                ex[j] = inner * k0;
            } else if (flat_rows) {
                u64 g = tile * T + t;
                have[j] = have[j] && g < total_rows_flat;
                spos[j] = g * R + n1;
            } else {
                u64 a = (dt * T + t) * Mrows + rest;                                   // row index, top digit = dt*T+t
                u64 kk0 = A.log_blow ? (a >> (A.log_n - A.log_R)) : 0;                 // virtual digit of this row
                u64 inner = (A.log_blow ? (a & ((n >> A.log_R) - 1)) : a) * R + n1;    // position inside the inner NTT
                spos[j] = A.src_is_coeffs ? (batch_idx * n + inner) : (batch_idx * ntot + a * R + n1);
                ex[j] = inner * kk0;
            }
            if (!have[j]) spos[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = Fe<F>::load(A.src + 4 * spos[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) if (!have[j]) x[j] = Fe<F>::zero();
    }
    if (A.first && A.log_blow) {
        Fe<F> w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = tw_get<F>(A.tw, A.log_ntot, ex[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) if (ex[j]) x[j] = mul<F>(x[j], w[j]);
    }
    __syncthreads();                                 // stage twiddles are in LDS

    // ---- the stages (decimation in frequency)
    if (odd) {
        if (active) {
            const u32 h = 1u << lh, sh = A.log_R - 1 - lh;
#pragma unroll
            for (int p = 0; p < 2; p++) {
                u32 i = (li[2 * p] >> A.log_T) & (h - 1);
                Fe<F> s = add<F>(x[2 * p], x[2 * p + 1]), d = sub<F>(x[2 * p], x[2 * p + 1]);
                if (i) d = mul<F>(d, lds_get<F>(twl, R / 2, i << sh));
                x[2 * p] = s; x[2 * p + 1] = d;
            }
        }
        lh -= 1;
    }
    bool fresh = !odd;                               // registers already hold the quad of the coming step
    while (lh >= 1) {
        const u32 q = 1u << (lh - 1);
        if (!fresh) {                                // through LDS to the quads of this step
            if (active) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (j < 2 ? ok01 : ok23) lds_put<F>(data, RT, li[j], x[j]);
            }
            __syncthreads();
            if (active) {
                u32 t = tid & (T - 1), pr = tid >> A.log_T;
                u32 i = pr & (q - 1), grp = pr >> (lh - 1);
                u32 lo = (grp << (lh + 1)) + i;
#pragma unroll
                for (int j = 0; j < 4; j++) { li[j] = (lo + j * q) * T + t; x[j] = lds_get<F>(data, RT, li[j]); }
            }
            ok01 = ok23 = active;
        }
        fresh = false;
        if (active) {
            const u32 i = (li[0] >> A.log_T) & (q - 1), sa_ = A.log_R - 1 - lh;
            // stage lh: pairs (0,2) with twiddle index i and (1,3) with i + q
            Fe<F> s0 = add<F>(x[0], x[2]), d0 = sub<F>(x[0], x[2]);
            Fe<F> s1 = add<F>(x[1], x[3]), d1 = sub<F>(x[1], x[3]);
            d1 = mul<F>(d1, lds_get<F>(twl, R / 2, (i + q) << sa_));
            if (i) d0 = mul<F>(d0, lds_get<F>(twl, R / 2, i << sa_));
            // stage lh-1: pairs (0,1) and (2,3), both with twiddle index i
            x[0] = add<F>(s0, s1); x[1] = sub<F>(s0, s1);
            x[2] = add<F>(d0, d1); x[3] = sub<F>(d0, d1);
            if (i) {
                Fe<F> wb = lds_get<F>(twl, R / 2, i << (sa_ + 1));
                x[1] = mul<F>(x[1], wb); x[3] = mul<F>(x[3], wb);
            }
        }
        lh -= 2;
    }

    // ---- write out (position n1 holds output bitrev(n1)); entry 0 of the scaled table is 1/N itself
    u64 dpos[4]; bool st[4];
    if (!row_mode) {
        u64 eo[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 n1 = li[j] >> A.log_T, t = li[j] & (T - 1);
            const u32 k1 = bitrev(n1, A.log_R);
            u64 b = bt * T + t;
            eo[j] = (k1 && b) ? ((u64)k1 * b) << (A.log_ntot - A.log_L) : 0;
            dpos[j] = batch_idx * ntot + k0 * n + sa * ((u64)1 << A.log_L) + (u64)k1 * B + b;
            st[j] = j < 2 ? ok01 : ok23;
        }
        Fe<F> w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = tw_get<F>(A.tw_out, A.log_ntot, eo[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) if (eo[j] || A.scale_out) x[j] = mul<F>(x[j], w[j]);
    } else {
        Fe<F> invn;
#pragma unroll
        for (int i = 0; i < 8; i++) invn.v[i] = A.inv_n[i];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 n1 = li[j] >> A.log_T, t = li[j] & (T - 1);
            const u32 k1 = bitrev(n1, A.log_R);
            st[j] = j < 2 ? ok01 : ok23;
            if (flat_rows) {
                u64 g = tile * T + t;
                st[j] = st[j] && g < total_rows_flat;
                dpos[j] = g * R + k1;
            } else {
                u64 a = (dt * T + t) * Mrows + rest;
                // digit-reverse the row index (digits most-significant first in A.dig)
                u64 rev = 0, wgt = 1;
                u32 shift = A.log_ntot - A.log_R;
                for (u32 d = 0; d < A.nd; d++) {
                    shift -= A.dig[d];
                    u64 digit = (a >> shift) & (((u64)1 << A.dig[d]) - 1);
                    rev += digit * wgt; wgt <<= A.dig[d];
                }
                dpos[j] = batch_idx * ntot + rev + ((u64)k1 << (A.log_ntot - A.log_R));
            }
            if (A.scale) x[j] = mul<F>(x[j], invn);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) if (st[j]) x[j].store(A.dst + 4 * dpos[j]);
}

// tab[e] *= c  (the inverse transform's inter-pass twiddles carry the 1/N)
template <class F>
__global__ void k_scale_table(u64* dst, const u64* src, const u32* c8, u64 count) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    Fe<F> c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.v[i] = c8[i];
    mul<F>(Fe<F>::load(src + 4 * e), c).store(dst + 4 * e);
}

// ---- twiddle table w^e, e < N/2, built on the device from the 2^i-th powers
template <class F>
__global__ void k_build_twiddles(u64* tw, const u64* pow2 /* w^(2^i), i < 32 */, u64 count) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    Fe<F> acc = Fe<F>::one();
    for (int i = 0; i < 32 && (e >> i); i++)
        if ((e >> i) & 1) acc = mul<F>(acc, Fe<F>::load(pow2 + 4 * i));
    acc.store(tw + 4 * e);
}

struct TwKey { int device; int field; unsigned logn; int inverse; bool operator<(const TwKey& o) const { return std::tie(device, field, logn, inverse) < std::tie(o.device, o.field, o.logn, o.inverse); } };
struct TwEntry { DevBuf tab; DevBuf tab_scaled /* tab * 1/N, built on first use by an inverse transform of two or more passes */; khost::fe inv_n; };
static std::map<TwKey, TwEntry> g_tw;          // twiddle tables per (device, field, size, direction); guarded by the device context's mutex
static std::mutex g_tw_mu;                      // ... and this one for the map itself (contexts of different devices share it)
void ntt_trim(Context& C) {                     // kh_trim: drop this device's tables (rebuilt on demand, ~20 us per table)
    std::lock_guard<std::mutex> lk(g_tw_mu);
    for (auto it = g_tw.begin(); it != g_tw.end();) { if (it->first.device == C.device) it = g_tw.erase(it); else ++it; }
}

khost::fe ntt_host_root(int field, unsigned logn, int inverse) {
    // w_{2^k} = (5^T)^(2^(32-k)); T = (p-1) >> 32   (kimchi/src/circuits/domains.rs:40-69)
    khost::Fld F(field);
    khost::fe five = {{5, 0, 0, 0}}; five = F.to_mont(five);
    khost::fe pm1 = F.f.p; pm1.l[0] -= 1;
    khost::fe Texp;
    for (int i = 0; i < 4; i++) Texp.l[i] = (pm1.l[i] >> 32) | (i < 3 ? pm1.l[i + 1] << 32 : 0);
    khost::fe acc = F.f.one, base = five;
    for (int i = 0; i < 256; i++) { if ((Texp.l[i >> 6] >> (i & 63)) & 1) acc = F.mul(acc, base); base = F.sqr(base); }
    for (unsigned i = logn; i < 32; i++) acc = F.sqr(acc);
    if (inverse) acc = F.inv(acc);
    return acc;
}

// w^e (or w^-e), e < max(n/2, 1), into a caller-provided device buffer of (n/2 + 32) x 32 bytes
int ntt_build_twiddles(Context& C, int field, unsigned logn, int inverse, u64* tab) {
    khost::Fld F(field);
    khost::fe pow2[32];
    pow2[0] = ntt_host_root(field, logn, inverse);
    for (int i = 1; i < 32; i++) pow2[i] = F.sqr(pow2[i - 1]);
    u64 count = logn ? ((u64)1 << (logn - 1)) : 1;
    u64* dpow = tab + count * 4;
    KH_HIP(hipMemcpyAsync(dpow, pow2, sizeof(pow2), hipMemcpyHostToDevice, C.stream));
    KH_HIP(hipStreamSynchronize(C.stream));       // pow2 is a stack buffer
    dim3 grid((unsigned)((count + 255) / 256));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_build_twiddles<FpParams>), grid, dim3(256), 0, C.stream, tab, dpow, count);
    else hipLaunchKernelGGL((k_build_twiddles<FqParams>), grid, dim3(256), 0, C.stream, tab, dpow, count);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

static int get_twiddles(Context& C, int field, unsigned logn, int inverse, TwEntry** out) {
    TwKey key{C.device, field, logn, inverse};
    std::lock_guard<std::mutex> lk(g_tw_mu);
    auto it = g_tw.find(key);
    if (it != g_tw.end()) { *out = &it->second; return KH_OK; }
    TwEntry& E = g_tw[key];
    khost::Fld F(field);
    u64 count = logn ? ((u64)1 << (logn - 1)) : 1;
    int rc = E.tab.reserve(count * 32 + 32 * 32); if (rc) { g_tw.erase(key); return rc; }
    if ((rc = ntt_build_twiddles(C, field, logn, inverse, E.tab.as<u64>()))) { g_tw.erase(key); return rc; }
    // the tables are per DEVICE, the streams per context: another context (kh_private_context_begin) may pick the entry up at once, so it is published
    // complete (once per (field, size, direction): ~20 us)
    if (hipStreamSynchronize(C.stream) != hipSuccess) { g_tw.erase(key); set_error("hipStreamSynchronize failed while building a twiddle table"); return KH_E_DEVICE; }
    khost::fe nn = {{(u64)1 << logn, 0, 0, 0}};
    E.inv_n = F.inv(F.to_mont(nn));
    *out = &E;
    return KH_OK;
}

// Largest sub-transform of a pass: KH_NTT_MAX_LOGR at start-up, kh_ntt_set_max_logr afterwards (4..10; 0 = back to the default).  The twiddle tables hold every
// power of the root, so they do not depend on the split and a change takes effect with the next transform.
static std::atomic<unsigned>& max_logr_cell() {
    static std::atomic<unsigned> v(getenv("KH_NTT_MAX_LOGR") ? (unsigned)std::min(10, std::max(4, atoi(getenv("KH_NTT_MAX_LOGR")))) : (unsigned)NTT_MAX_LOGR);
    return v;
}
unsigned ntt_max_logr() { return max_logr_cell().load(std::memory_order_relaxed); }
int ntt_set_max_logr(unsigned v) {
    if (v == 0) v = NTT_MAX_LOGR;
    if (v < 4 || v > 10) { set_error("kh_ntt_set_max_logr: %u is outside 4..10", v); return KH_E_INVALID; }
    max_logr_cell().store(v, std::memory_order_relaxed);
    return KH_OK;
}

// split log_n into passes of at most ntt_max_logr() bits, most significant (first pass) first
static std::vector<unsigned> split_passes(unsigned log_n) {
    std::vector<unsigned> r;
    if (log_n == 0) return r;
    const unsigned max_logr = ntt_max_logr();
    unsigned P = (log_n + max_logr - 1) / max_logr;
    unsigned base = log_n / P, extra = log_n % P;
    for (unsigned i = 0; i < P; i++) r.push_back(base + (i < extra ? 1 : 0));
    return r;
}

// (The passes on nine 29-bit limbs -- built, bit-exact and measured SLOWER in round 4: iNTT 2^16 x 19 0.122 against 0.115 ms, the 16-column extension
// 0.808 against 0.750 ms, profiles/r04_ntt29_* -- live in tools/ntt29/ with their limb model; DESIGN.md section 4 has the analysis.)
template <class F>
static int launch_pass(Context& C, const PassArgs& A, u64 tiles) {
    static const size_t lds_pad = getenv("KH_NTT_LDS_PAD") ? (size_t)atol(getenv("KH_NTT_LDS_PAD")) : 0;      // (occupancy experiments: extra dynamic LDS per workgroup)
    size_t lds = ((size_t)4 << (A.log_R + A.log_T)) * 8 + ((size_t)4 << (A.log_R ? A.log_R - 1 : 0)) * 8 + lds_pad;
    hipLaunchKernelGGL((k_ntt_pass<F, NTT_THREADS>), dim3((unsigned)tiles), dim3(NTT_THREADS), lds, C.stream, A);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

static int get_scaled_table(Context& C, int field, unsigned log_ntot, TwEntry* E, const u64** out) {
    u64 count = log_ntot ? ((u64)1 << (log_ntot - 1)) : 1;
    std::lock_guard<std::mutex> lk(g_tw_mu);            // (callers of different contexts share the entry)
    if (!E->tab_scaled.p) {
        int rc = E->tab_scaled.reserve(count * 32 + 32); if (rc) return rc;
        u32* c8 = (u32*)(E->tab_scaled.as<u64>() + count * 4);
        KH_HIP(hipMemcpyAsync(c8, &E->inv_n, 32, hipMemcpyHostToDevice, C.stream));      // inv_n lives in the table entry
        dim3 grid((unsigned)((count + 255) / 256));
        if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_scale_table<FpParams>), grid, dim3(256), 0, C.stream, E->tab_scaled.as<u64>(), E->tab.as<u64>(), c8, count);
        else hipLaunchKernelGGL((k_scale_table<FqParams>), grid, dim3(256), 0, C.stream, E->tab_scaled.as<u64>(), E->tab.as<u64>(), c8, count);
        KH_HIP(hipGetLastError());
        KH_HIP(hipStreamSynchronize(C.stream));         // published complete, as the table itself
    }
    *out = E->tab_scaled.as<u64>();
    return KH_OK;
}

// Generic driver: `src` holds batch x n coefficients/evaluations; the result (batch x n*2^log_blow,
// natural order) is written to `dst`; `tmp` (same size as dst) is scratch.  src may equal dst
// when log_blow == 0.
template <class F>
static int transform(Context& C, int field, const u64* src, u64* dst, u64* tmp, unsigned log_n, unsigned log_blow,
                     int inverse, size_t batch) {
    const unsigned log_ntot = log_n + log_blow;
    TwEntry* E; int rc = get_twiddles(C, field, log_ntot, inverse, &E); if (rc) return rc;
    std::vector<unsigned> passes = split_passes(log_n);
    const size_t P = passes.size();
    PassArgs A; memset(&A, 0, sizeof(A));
    A.tw = E->tab.as<u64>(); A.tw_out = A.tw; A.log_ntot = log_ntot; A.log_n = log_n; A.log_blow = log_blow; A.batch = batch;
    memcpy(A.inv_n, &E->inv_n, 32);
    if (P == 0) {
        // n == 1: each output of the extension equals the single coefficient; plain NTT of size 1 is the identity
        if (log_blow == 0) { if (src != dst) KH_HIP(hipMemcpyAsync(dst, src, batch * 32, hipMemcpyDeviceToDevice, C.stream)); return KH_OK; }
        set_error("kh_lde with n == 1 is not supported"); return KH_E_INVALID;
    }
    // An inverse transform of two or more passes folds the 1/N into the first pass's inter-pass twiddles (one product per
    // element less); a single pass scales on the way out.
    const u64* scaled = nullptr;
    if (inverse && P >= 2 && (rc = get_scaled_table(C, field, log_ntot, E, &scaled))) return rc;
    // buffer plan: pass 1 reads src; middle passes run in place on tmp; the last pass writes dst.
    unsigned log_L = log_n;
    for (size_t p = 0; p < P; p++) {
        const bool first = p == 0, last = p + 1 == P;
        A.log_R = passes[p]; A.log_L = log_L;
        A.first = first; A.last = last;
        A.src_is_coeffs = (first && log_blow) ? 1 : 0;
        A.scale = (last && inverse && !scaled) ? 1 : 0;
        A.scale_out = (first && scaled) ? 1 : 0;
        A.tw_out = (first && scaled) ? scaled : A.tw;
        A.src = first ? src : tmp;
        A.dst = last ? dst : tmp;
        unsigned log_T = NTT_LOG_TILE - A.log_R;
        u64 tiles;
        if (!last) {
            unsigned log_B = log_L - A.log_R;
            if (log_T > log_B) log_T = log_B;
            A.log_T = log_T;
            tiles = (u64)batch << (log_ntot - A.log_R - log_T);
        } else {
            A.nd = 0;
            if (log_blow) A.dig[A.nd++] = log_blow;
            for (size_t q = 0; q + 1 < P; q++) A.dig[A.nd++] = passes[q];
            if (A.nd == 0) {          // single pass, tiles run across the batch
                u64 lt = 0; while (((u64)1 << (lt + 1)) <= batch && lt + 1 <= log_T) lt++;
                A.log_T = (u32)lt;
                tiles = (batch + ((u64)1 << lt) - 1) >> lt;
            } else {
                if (log_T > A.dig[0]) log_T = A.dig[0];
                A.log_T = log_T;
                tiles = (u64)batch << (log_ntot - A.log_R - log_T);
            }
        }
        if ((rc = launch_pass<F>(C, A, tiles))) return rc;
        log_L -= A.log_R;
    }
    return KH_OK;
}

int ntt_run(Context& C, int field, uint64_t* data_dev, unsigned log2_n, int inverse, size_t batch) {
    size_t bytes = (batch << log2_n) * 32;
    int rc = C.ws_ntt_b.reserve(bytes); if (rc) return rc;
    C.timer.begin(C.stream);
    if (field == KH_FIELD_FP) rc = transform<FpParams>(C, field, data_dev, data_dev, C.ws_ntt_b.as<u64>(), log2_n, 0, inverse, batch);
    else rc = transform<FqParams>(C, field, data_dev, data_dev, C.ws_ntt_b.as<u64>(), log2_n, 0, inverse, batch);
    C.timer.mark(inverse ? "intt" : "ntt", C.stream);
    return rc;
}
int lde_run(Context& C, int field, const uint64_t* coeffs_dev, unsigned log2_n, unsigned log2_blowup, uint64_t* out_dev, size_t batch) {
    if (log2_blowup == 0) {
        size_t bytes = (batch << log2_n) * 32;
        if (coeffs_dev != out_dev) KH_HIP(hipMemcpyAsync(out_dev, coeffs_dev, bytes, hipMemcpyDeviceToDevice, C.stream));
        return ntt_run(C, field, out_dev, log2_n, 0, batch);
    }
    size_t bytes = (batch << (log2_n + log2_blowup)) * 32;
    DevBuf& lde_tmp = C.scratch("lde_tmp");
    int rc = lde_tmp.reserve(bytes); if (rc) return rc;
    C.timer.begin(C.stream);
    if (field == KH_FIELD_FP) rc = transform<FpParams>(C, field, coeffs_dev, out_dev, lde_tmp.as<u64>(), log2_n, log2_blowup, 0, batch);
    else rc = transform<FqParams>(C, field, coeffs_dev, out_dev, lde_tmp.as<u64>(), log2_n, log2_blowup, 0, batch);
    C.timer.mark("lde", C.stream);
    return rc;
}

}  // namespace kh
