// ipa.hip -- the per-round vector operations of the IPA prover (poly-commitment/src/ipa.rs:929-1007),
// what stays on the CPU between the L/R MSMs of consecutive rounds.
//   fold_scalars : a' = a_lo + u^-1 * a_hi  /  b' = b_lo + u * b_hi          (ipa.rs:980-1003)
//   fold_points  : g' = g_lo + [u] g_hi  = CommitmentCurve::combine_one       (ipa.rs:1006, commitment.rs:576-579)
//   fold_points_endo : the same by the endo ladder of combine_one_endo       (combine.rs:292-340)
//   round_prepare / round_fold : the device-resident opening loop (no basis folding, see below)
//   inner_product: <a, b>                                                      (utils/src/field_helpers.rs:273-279)
// Element-wise and embarrassingly parallel; the basis fold is one 255-bit double-and-add per point
// (a ~380-operation dependent chain: ~4 ms whatever the length below ~60k points).
#include "common.hpp"
#include "curve.cuh"
#include "host_ec.hpp"
#include "msm.hpp"

namespace kh {

template <class F>
__global__ void k_fold_scalars(const u64* __restrict__ lo, const u64* __restrict__ hi, const u64* __restrict__ u, size_t n, u64* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> U = Fe<F>::load(u);
    add<F>(Fe<F>::load(lo + 4 * i), mul<F>(U, Fe<F>::load(hi + 4 * i))).store(out + 4 * i);
}
// per-block partial sums of a_i * b_i (field addition is associative: any order gives the same element)
template <class F>
__global__ void __launch_bounds__(256)
k_inner_product(const u64* __restrict__ a, const u64* __restrict__ b, size_t n, u64* __restrict__ partial) {
    __shared__ u32 sh[256 * 8];
    Fe<F> acc = Fe<F>::zero();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc = add<F>(acc, mul<F>(Fe<F>::load(a + 4 * i), Fe<F>::load(b + 4 * i)));
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k * 256 + threadIdx.x] = acc.v[k];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
            Fe<F> o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[k * 256 + threadIdx.x + s];
            acc = add<F>(acc, o);
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k * 256 + threadIdx.x] = acc.v[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) acc.store(partial + 4 * blockIdx.x);
}
template <class BF>
__global__ void __launch_bounds__(128)
k_fold_points(const uint8_t* __restrict__ g_lo, const uint8_t* __restrict__ g_hi, const u64* __restrict__ u_plain, size_t n,
              uint8_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 kw[8];
#pragma unroll
    for (int w = 0; w < 4; w++) { u64 l = u_plain[w]; kw[2 * w] = (u32)l; kw[2 * w + 1] = (u32)(l >> 32); }
    Xyzz<BF> v = scalar_mul<BF>(Xyzz<BF>::from_affine(Aff<BF>::load(g_hi + i * 64)), kw);
    v = madd<BF>(v, Aff<BF>::load(g_lo + i * 64), false);
    Fe<BF> x = Fe<BF>::zero(), y = Fe<BF>::zero();
    uint8_t inf = 1;
    if (!v.is_identity()) {
        Fe<BF> izzz = inv<BF>(v.zzz);
        Fe<BF> izz = sqr<BF>(mul<BF>(izzz, v.zz));
        x = mul<BF>(v.x, izz); y = mul<BF>(v.y, izzz);
        inf = 0;
    }
    x.store(out_xy + i * 64); y.store(out_xy + i * 64 + 32);
    out_inf[i] = inf;
}

// The endo ladder of CommitmentCurve::combine_one_endo (commitment.rs:581-589 -> combine.rs:292-340,
// Halo section 6.2): acc = 2 (phi(g2) + g2); for the 64 two-bit chunks of the 128-bit challenge, high to low:
// s = +-g2 (bit 2i), phi(s) if bit 2i+1; acc = (acc + s) + acc; result g1 + acc = g1 + [to_field(chal)] g2.
// 64 x (doubling + mixed addition) instead of a 255-bit double-and-add: 3.4x fewer field products.
template <class BF>
__global__ void __launch_bounds__(128)
k_fold_points_endo(const uint8_t* __restrict__ g_lo, const uint8_t* __restrict__ g_hi, u64 chal_lo, u64 chal_hi,
                   const u64* __restrict__ endo_q, size_t n, uint8_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe<BF> eq = Fe<BF>::load(endo_q);
    const Aff<BF> P2 = Aff<BF>::load(g_hi + i * 64);
    Aff<BF> phi = P2; phi.x = mul<BF>(P2.x, eq);
    Xyzz<BF> acc = dbl<BF>(madd<BF>(Xyzz<BF>::from_affine(phi), P2, false));
    for (int k = 63; k >= 0; k--) {
        u64 w = k >= 32 ? chal_hi : chal_lo;
        int sh = 2 * (k & 31);
        bool b0 = (w >> sh) & 1ull, b1 = (w >> (sh + 1)) & 1ull;
        Aff<BF> S = P2;
        if (b1) S.x = phi.x;
        acc = madd<BF>(dbl<BF>(acc), S, !b0);              // (acc + s) + acc = 2 acc + s
    }
    acc = madd<BF>(acc, Aff<BF>::load(g_lo + i * 64), false);
    Fe<BF> x = Fe<BF>::zero(), y = Fe<BF>::zero();
    uint8_t inf = 1;
    if (!acc.is_identity()) {
        Fe<BF> izzz = inv<BF>(acc.zzz);
        Fe<BF> izz = sqr<BF>(mul<BF>(izzz, acc.zz));
        x = mul<BF>(acc.x, izz); y = mul<BF>(acc.y, izzz);
        inf = 0;
    }
    x.store(out_xy + i * 64); y.store(out_xy + i * 64 + 32);
    out_inf[i] = inf;
}

// ---------------------------------------------------------------- device-resident opening rounds
// The folding loop of SRS::open (ipa.rs:929-1007) without ever folding the basis.  After j rounds the folded
// basis is  g_j[i] = sum_{t = i mod N_j} coef_j[t div N_j] * G_t  (N_j = N / 2^j, coef_j = the tensor of the
// challenges so far), so  L_j = <a_hi, g_j,lo> + [rand_l] H + [<a_hi, b_lo>] U  is ONE MSM over the ORIGINAL
// basis (whose window tables are already resident) with the expanded scalars  a_hi[t mod N_j - 0] * coef_j[t div N_j]
// on the points of the low halves and zero elsewhere; R_j likewise on the high halves.  A size-N MSM per round is
// throughput work (the path this library is fastest at); folding N_j points by the endo ladder is a ~1400-product
// dependent chain per point, i.e. ~0.9 ms of latency per round however few points are left.  The final basis
// element sg = g_16[0] = <coef_16, G> is one more MSM.  Group elements are equal to the reference's, hence
// bit-identical after normalisation.
struct Fe4 { u64 l[4]; };

// side 0 (L): <a[m..2m), b[0..m)>; side 1 (R): <a[0..m), b[m..2m)>; per-block partial sums
template <class F>
__global__ void __launch_bounds__(256)
k_ipa_ip(const u64* __restrict__ a, const u64* __restrict__ b, size_t m, u64* __restrict__ partial) {
    __shared__ u32 sh[256 * 8];
    const int side = blockIdx.y;
    const u64* pa = side == 0 ? a + 4 * m : a;
    const u64* pb = side == 0 ? b : b + 4 * m;
    Fe<F> acc = Fe<F>::zero();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x)
        acc = add<F>(acc, mul<F>(Fe<F>::load(pa + 4 * i), Fe<F>::load(pb + 4 * i)));
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k * 256 + threadIdx.x] = acc.v[k];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
            Fe<F> o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[k * 256 + threadIdx.x + s];
            acc = add<F>(acc, o);
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k * 256 + threadIdx.x] = acc.v[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) acc.store(partial + 4 * ((size_t)side * gridDim.x + blockIdx.x));
}
// block = side: sums the <= 64 partials and fills the two extra scalar slots of that side's MSM:
// sc[side][n] = rand (the H term), sc[side][n + 1] = the inner product (the U term)
template <class F>
__global__ void __launch_bounds__(64)
k_ipa_ip_fin(const u64* __restrict__ partial, unsigned nblk, Fe4 rand_l, Fe4 rand_r, size_t n, u64* __restrict__ sc) {
    __shared__ u32 sh[64 * 8];
    const int side = blockIdx.x;
    Fe<F> acc = Fe<F>::zero();
    if (threadIdx.x < nblk) acc = Fe<F>::load(partial + 4 * ((size_t)side * nblk + threadIdx.x));
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k * 64 + threadIdx.x] = acc.v[k];
    __syncthreads();
    for (int s = 32; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
            Fe<F> o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = sh[k * 64 + threadIdx.x + s];
            acc = add<F>(acc, o);
#pragma unroll
            for (int k = 0; k < 8; k++) sh[k * 64 + threadIdx.x] = acc.v[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        u64* dst = sc + 4 * ((size_t)side * (n + 2) + n);
        const Fe4& r = side == 0 ? rand_l : rand_r;
        dst[0] = r.l[0]; dst[1] = r.l[1]; dst[2] = r.l[2]; dst[3] = r.l[3];
        acc.store(dst + 4);
    }
}
// the two expanded scalar vectors over the original basis (N_j = 2m = 2^logN)
template <class F>
__global__ void k_ipa_expand(const u64* __restrict__ a, const u64* __restrict__ coef, size_t n, size_t m, unsigned logN, u64* __restrict__ sc) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const size_t r = t & (2 * m - 1), q = t >> logN;
    const bool lo = r < m;
    Fe<F> v = mul<F>(Fe<F>::load(a + 4 * (lo ? r + m : r - m)), Fe<F>::load(coef + 4 * q));
    const Fe<F> z = Fe<F>::zero();
    (lo ? v : z).store(sc + 4 * t);
    (lo ? z : v).store(sc + 4 * ((n + 2) + t));
}
// a' = a_lo + u^-1 a_hi, b' = b_lo + u b_hi (ipa.rs:980-1003); coef' = coef (x) (1, u)  (combine_one_endo's scalar, ipa.rs:1006)
template <class F>
__global__ void k_ipa_fold(const u64* __restrict__ a, const u64* __restrict__ b, const u64* __restrict__ coef, size_t m, size_t ncoef,
                           Fe4 u4, Fe4 uinv4, u64* __restrict__ a2, u64* __restrict__ b2, u64* __restrict__ coef2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const Fe<F> u = Fe<F>::load(u4.l);
    if (i < m) {
        const Fe<F> ui = Fe<F>::load(uinv4.l);
        add<F>(Fe<F>::load(a + 4 * i), mul<F>(ui, Fe<F>::load(a + 4 * (i + m)))).store(a2 + 4 * i);
        add<F>(Fe<F>::load(b + 4 * i), mul<F>(u, Fe<F>::load(b + 4 * (i + m)))).store(b2 + 4 * i);
    }
    if (i < ncoef) {
        const Fe<F> c = Fe<F>::load(coef + 4 * i);
        c.store(coef2 + 8 * i);
        mul<F>(c, u).store(coef2 + 8 * i + 4);
    }
}

// One launch per round: the PENDING fold of the previous round (a' = a_lo + u^-1 a_hi, b' = b_lo + u b_hi, coef' = coef (x) (1, u);
// kh_ipa_round_fold only records u), the two inner products of this round over the folded vectors, and the expanded scalars.
// Thread t expands point t (recomputing the folded a' it needs: a product, not a dependency on another thread's store); threads
// below m = N'/2 also materialise a', b' at i and i + m and accumulate <a'_hi, b'_lo> / <a'_lo, b'_hi>; the block sums go to
// `partial`, and the LAST block to finish (agent-scope counter) adds them up and fills the H / U scalar slots of both sides.
// Replaces k_ipa_fold + k_ipa_ip + k_ipa_ip_fin + k_ipa_expand: four dependent launches (~25 us of an opening round) -> one.
template <class F>
__global__ void __launch_bounds__(256)
k_ipa_step(const u64* __restrict__ a, const u64* __restrict__ b, const u64* __restrict__ coef, size_t n, size_t cur2, unsigned logN2, size_t ncoef,
           int has_fold, Fe4 u4, Fe4 ui4, u64* __restrict__ a2, u64* __restrict__ b2, u64* __restrict__ coef2,
           Fe4 rand_l, Fe4 rand_r, u64* __restrict__ sc, u64* __restrict__ partial, unsigned* __restrict__ counter) {
    __shared__ u32 sh[8 * 8];
    __shared__ unsigned is_last;
    __builtin_amdgcn_s_setprio(3);                         // the head of a round's dependent chain: above whatever throughput work shares the CUs (msm.hip, KH_HIGH_PRIO)
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t m = cur2 / 2;
    const Fe<F> u = Fe<F>::load(u4.l), ui = Fe<F>::load(ui4.l);
    auto a_at = [&](size_t x) { Fe<F> v = Fe<F>::load(a + 4 * x); if (has_fold) v = add<F>(v, mul<F>(ui, Fe<F>::load(a + 4 * (x + cur2)))); return v; };
    auto b_at = [&](size_t x) { Fe<F> v = Fe<F>::load(b + 4 * x); if (has_fold) v = add<F>(v, mul<F>(u, Fe<F>::load(b + 4 * (x + cur2)))); return v; };
    if (t < n) {                                           // expand (ipa.rs:943-961 over the original basis, see above)
        const size_t r = t & (cur2 - 1), q = t >> logN2;
        const bool lo = r < m;
        Fe<F> c = Fe<F>::load(coef + 4 * (has_fold ? q >> 1 : q));
        if (has_fold && (q & 1)) c = mul<F>(c, u);
        const Fe<F> v = mul<F>(a_at(lo ? r + m : r - m), c), z = Fe<F>::zero();
        (lo ? v : z).store(sc + 4 * t);
        (lo ? z : v).store(sc + 4 * ((n + 2) + t));
    }
    if (has_fold && t < 2 * ncoef) {                       // coef' = coef (x) (1, u)
        Fe<F> c = Fe<F>::load(coef + 4 * (t >> 1));
        if (t & 1) c = mul<F>(c, u);
        c.store(coef2 + 4 * t);
    }
    Fe<F> accL = Fe<F>::zero(), accR = Fe<F>::zero();
    const size_t nblk_ip = (m + blockDim.x - 1) / blockDim.x;
    if (t < m) {
        const Fe<F> alo = a_at(t), ahi = a_at(t + m), blo = b_at(t), bhi = b_at(t + m);
        if (has_fold) { alo.store(a2 + 4 * t); ahi.store(a2 + 4 * (t + m)); blo.store(b2 + 4 * t); bhi.store(b2 + 4 * (t + m)); }
        accL = mul<F>(ahi, blo); accR = mul<F>(alo, bhi);
    }
    auto wave_sum = [](Fe<F> v) {                          // sum over the 64 lanes of a wave (valid in lane 0)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            Fe<F> o;
#pragma unroll
            for (int k = 0; k < 8; k++) o.v[k] = __shfl_down(v.v[k], d, 64);
            v = add<F>(v, o);
        }
        return v;
    };
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (blockIdx.x < nblk_ip) {                            // block sums of the two inner products: wave shuffles, then four values through LDS
        accL = wave_sum(accL); accR = wave_sum(accR);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) { sh[(wave * 2) * 8 + k] = accL.v[k]; sh[(wave * 2 + 1) * 8 + k] = accR.v[k]; }
        }
        __syncthreads();
        if (threadIdx.x < 2) {                             // thread = side
            Fe<F> acc = Fe<F>::zero();
            for (u32 w = 0; w < 4; w++) {
                Fe<F> o;
#pragma unroll
                for (int k = 0; k < 8; k++) o.v[k] = sh[(w * 2 + threadIdx.x) * 8 + k];
                acc = add<F>(acc, o);
            }
            acc.store(partial + 4 * ((size_t)threadIdx.x * nblk_ip + blockIdx.x));
        }
    }
    // last block out: the final sums and the two extra scalar slots of each side (H: the blinder, U: the inner product)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x < nblk_ip) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // only these blocks publish something the last block reads
        is_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    {
        const u32 side = threadIdx.x >> 7, x = threadIdx.x & 127u;        // waves 0-1: L, waves 2-3: R
        Fe<F> acc = Fe<F>::zero();
        for (size_t k = x; k < nblk_ip; k += 128) {
            const u64* src = partial + 4 * ((size_t)side * nblk_ip + k);
            u64 l[4];
#pragma unroll
            for (int w = 0; w < 4; w++) l[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // written by other blocks
            acc = add<F>(acc, Fe<F>::load(l));
        }
        acc = wave_sum(acc);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) sh[wave * 8 + k] = acc.v[k];
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            Fe<F> o, r2;
#pragma unroll
            for (int k = 0; k < 8; k++) { o.v[k] = sh[(2 * threadIdx.x) * 8 + k]; r2.v[k] = sh[(2 * threadIdx.x + 1) * 8 + k]; }
            o = add<F>(o, r2);
            u64* dst = sc + 4 * ((size_t)threadIdx.x * (n + 2) + n);
            const Fe4& r = threadIdx.x == 0 ? rand_l : rand_r;
            dst[0] = r.l[0]; dst[1] = r.l[1]; dst[2] = r.l[2]; dst[3] = r.l[3];
            o.store(dst + 4);
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-armed for the next round
}
// cur2 = the vector length AFTER the pending fold (= cur if none); a2 / b2 / coef2 receive the folded vectors
int ipa_round_step(hipStream_t s, int field, int has_fold, const uint64_t* a, const uint64_t* b, const uint64_t* coef, size_t n, size_t cur2, size_t ncoef,
                   const uint64_t u[4], const uint64_t uinv[4], uint64_t* a2, uint64_t* b2, uint64_t* coef2,
                   const uint64_t rand_l[4], const uint64_t rand_r[4], uint64_t* sc, uint64_t* partial, unsigned* counter) {
    unsigned logN2 = 0; while (((size_t)1 << logN2) < cur2) logN2++;
    Fe4 rl, rr, u4, ui4; memcpy(rl.l, rand_l, 32); memcpy(rr.l, rand_r, 32); memcpy(u4.l, u, 32); memcpy(ui4.l, uinv, 32);
    dim3 g((unsigned)((n + 255) / 256));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_ipa_step<FpParams>), g, dim3(256), 0, s, a, b, coef, n, cur2, logN2, ncoef, has_fold, u4, ui4, a2, b2, coef2, rl, rr, sc, partial, counter);
    else hipLaunchKernelGGL((k_ipa_step<FqParams>), g, dim3(256), 0, s, a, b, coef, n, cur2, logN2, ncoef, has_fold, u4, ui4, a2, b2, coef2, rl, rr, sc, partial, counter);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

int ipa_round_prepare(hipStream_t s, int field, const uint64_t* a, const uint64_t* b, const uint64_t* coef, size_t n, size_t Nj,
                      const uint64_t rand_l[4], const uint64_t rand_r[4], uint64_t* sc, uint64_t* partial) {
    const size_t m = Nj / 2;
    unsigned logN = 0; while (((size_t)1 << logN) < Nj) logN++;
    const unsigned nblk = (unsigned)std::min<size_t>(64, (m + 255) / 256);
    Fe4 rl, rr; memcpy(rl.l, rand_l, 32); memcpy(rr.l, rand_r, 32);
    dim3 eg((unsigned)((n + 255) / 256));
    if (field == KH_FIELD_FP) {
        hipLaunchKernelGGL((k_ipa_ip<FpParams>), dim3(nblk, 2), dim3(256), 0, s, a, b, m, partial);
        hipLaunchKernelGGL((k_ipa_ip_fin<FpParams>), dim3(2), dim3(64), 0, s, partial, nblk, rl, rr, n, sc);
        hipLaunchKernelGGL((k_ipa_expand<FpParams>), eg, dim3(256), 0, s, a, coef, n, m, logN, sc);
    } else {
        hipLaunchKernelGGL((k_ipa_ip<FqParams>), dim3(nblk, 2), dim3(256), 0, s, a, b, m, partial);
        hipLaunchKernelGGL((k_ipa_ip_fin<FqParams>), dim3(2), dim3(64), 0, s, partial, nblk, rl, rr, n, sc);
        hipLaunchKernelGGL((k_ipa_expand<FqParams>), eg, dim3(256), 0, s, a, coef, n, m, logN, sc);
    }
    KH_HIP(hipGetLastError());
    return KH_OK;
}
int ipa_round_fold(hipStream_t s, int field, const uint64_t* a, const uint64_t* b, const uint64_t* coef, size_t Nj, size_t ncoef,
                   const uint64_t u[4], const uint64_t uinv[4], uint64_t* a2, uint64_t* b2, uint64_t* coef2) {
    const size_t m = Nj / 2, thr = std::max(m, ncoef);
    Fe4 u4, ui4; memcpy(u4.l, u, 32); memcpy(ui4.l, uinv, 32);
    dim3 g((unsigned)((thr + 255) / 256));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_ipa_fold<FpParams>), g, dim3(256), 0, s, a, b, coef, m, ncoef, u4, ui4, a2, b2, coef2);
    else hipLaunchKernelGGL((k_ipa_fold<FqParams>), g, dim3(256), 0, s, a, b, coef, m, ncoef, u4, ui4, a2, b2, coef2);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

// sg = <coef_R, G> split on the LAST challenge: coef_R[2s + b] = coef_{R-1}[s] u_R^b, so sg = A + u_R B with A = sum_s coef_{R-1}[s] G_{2s} and
// B = sum_s coef_{R-1}[s] G_{2s+1} -- two MSMs that need only the first R-1 challenges and therefore run on a side slot DURING the last round
// (kh_ipa_open); the host finishes with one scalar multiplication.  out = [A scalars | B scalars], n each; coef holds coef_{R-2} when the
// fold of round R-1 is still pending (has_fold: u = its challenge), else coef_{R-1}.
template <class F>
__global__ void k_sg_split(const u64* __restrict__ coef, size_t n, int has_fold, Fe4 u4, u64* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const size_t s = t >> 1;
    Fe<F> c = Fe<F>::load(coef + 4 * (has_fold ? s >> 1 : s));
    if (has_fold && (s & 1)) c = mul<F>(c, Fe<F>::load(u4.l));
    const Fe<F> z = Fe<F>::zero();
    ((t & 1) ? z : c).store(out + 4 * t);
    ((t & 1) ? c : z).store(out + 4 * (n + t));
}
int ipa_sg_split(hipStream_t s, int field, const uint64_t* coef, size_t n, int has_fold, const uint64_t u[4], uint64_t* out) {
    Fe4 u4; memcpy(u4.l, u, 32);
    dim3 g((unsigned)((n + 255) / 256));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_sg_split<FpParams>), g, dim3(256), 0, s, coef, n, has_fold, u4, out);
    else hipLaunchKernelGGL((k_sg_split<FqParams>), g, dim3(256), 0, s, coef, n, has_fold, u4, out);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

// ---------------------------------------------------------------- challenge polynomial coefficients
// b_poly_coefficients (commitment.rs:464-476): s[i] = prod_{j : bit j of i} chals[rounds - 1 - j].  One thread per
// coefficient, <= rounds products.  With `rs`: out[i] = sum_j rs[j] * s_j[i] over the k challenge sets, the
// weighted sum of challenge polynomials that batch_dlog_accumulator_check (utils.rs:212-273, weights -r^j) and the
// batch verifier (ipa.rs:402-420, weights sg_rand_base^j) multiply into the SRS.
template <class F>
__global__ void k_bpoly(const u64* __restrict__ chals, unsigned rounds, size_t k, const u64* __restrict__ rs, u64* __restrict__ out) {
    const size_t len = (size_t)1 << rounds;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    Fe<F> acc = Fe<F>::zero();
    for (size_t j = 0; j < k; j++) {
        Fe<F> prod = rs ? Fe<F>::load(rs + 4 * j) : Fe<F>::one();
        for (unsigned b = 0; b < rounds; b++)
            if ((i >> b) & 1) prod = mul<F>(prod, Fe<F>::load(chals + 4 * (j * rounds + (rounds - 1 - b))));
        if (rs) acc = add<F>(acc, prod);
        else prod.store(out + 4 * (j * len + i));
    }
    if (rs) acc.store(out + 4 * i);
}
int bpoly_run(hipStream_t s, int field, const uint64_t* chals_dev, unsigned rounds, size_t k, const uint64_t* rs_dev, uint64_t* out_dev) {
    dim3 g((unsigned)((((size_t)1 << rounds) + 255) / 256));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_bpoly<FpParams>), g, dim3(256), 0, s, chals_dev, rounds, k, rs_dev, out_dev);
    else hipLaunchKernelGGL((k_bpoly<FqParams>), g, dim3(256), 0, s, chals_dev, rounds, k, rs_dev, out_dev);
    KH_HIP(hipGetLastError());
    return KH_OK;
}

#define g_ipa_a (kh::ctx().scratch("ipa_a"))
#define g_ipa_b (kh::ctx().scratch("ipa_b"))
#define g_ipa_c (kh::ctx().scratch("ipa_c"))

int ipa_fold_scalars(Context& C, int field, const uint64_t* lo, const uint64_t* hi, const uint64_t u[4], size_t n, uint64_t* out) {
    int rc;
    if ((rc = g_ipa_a.reserve(n * 64 + 32))) return rc;
    if ((rc = g_ipa_b.reserve(n * 32))) return rc;
    u64* dlo = g_ipa_a.as<u64>(); u64* dhi = dlo + 4 * n; u64* du = dhi + 4 * n;
    hipStream_t s = C.stream;
    KH_HIP(hipMemcpyAsync(dlo, lo, n * 32, hipMemcpyHostToDevice, s));
    KH_HIP(hipMemcpyAsync(dhi, hi, n * 32, hipMemcpyHostToDevice, s));
    KH_HIP(hipMemcpyAsync(du, u, 32, hipMemcpyHostToDevice, s));
    dim3 grid((unsigned)((n + 255) / 256));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_fold_scalars<FpParams>), grid, dim3(256), 0, s, dlo, dhi, du, n, g_ipa_b.as<u64>());
    else hipLaunchKernelGGL((k_fold_scalars<FqParams>), grid, dim3(256), 0, s, dlo, dhi, du, n, g_ipa_b.as<u64>());
    KH_HIP(hipGetLastError());
    KH_HIP(hipMemcpyAsync(out, g_ipa_b.p, n * 32, hipMemcpyDeviceToHost, s));
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}
int ipa_inner_product(Context& C, int field, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
    int rc;
    const unsigned blocks = (unsigned)std::min<size_t>(256, (n + 255) / 256);
    if ((rc = g_ipa_a.reserve(n * 64))) return rc;
    if ((rc = g_ipa_b.reserve((size_t)blocks * 32))) return rc;
    u64* da = g_ipa_a.as<u64>(); u64* db = da + 4 * n;
    hipStream_t s = C.stream;
    KH_HIP(hipMemcpyAsync(da, a, n * 32, hipMemcpyHostToDevice, s));
    KH_HIP(hipMemcpyAsync(db, b, n * 32, hipMemcpyHostToDevice, s));
    if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_inner_product<FpParams>), dim3(blocks), dim3(256), 0, s, da, db, n, g_ipa_b.as<u64>());
    else hipLaunchKernelGGL((k_inner_product<FqParams>), dim3(blocks), dim3(256), 0, s, da, db, n, g_ipa_b.as<u64>());
    KH_HIP(hipGetLastError());
    std::vector<khost::fe> part(blocks);
    KH_HIP(hipMemcpyAsync(part.data(), g_ipa_b.p, (size_t)blocks * 32, hipMemcpyDeviceToHost, s));
    KH_HIP(hipStreamSynchronize(s));
    khost::Fld F(field);
    khost::fe acc; memset(&acc, 0, sizeof(acc));
    for (unsigned i = 0; i < blocks; i++) acc = F.add(acc, part[i]);
    memcpy(out, &acc, 32);
    return KH_OK;
}
int ipa_fold_points(Context& C, int curve, const uint64_t* g_lo, const uint64_t* g_hi, const uint64_t u[4], size_t n,
                    uint64_t* out_xy, uint8_t* out_inf) {
    int rc;
    if ((rc = g_ipa_a.reserve(n * 128 + 32))) return rc;
    if ((rc = g_ipa_b.reserve(n * 64))) return rc;
    if ((rc = g_ipa_c.reserve(n))) return rc;
    uint8_t* dlo = g_ipa_a.as<uint8_t>(); uint8_t* dhi = dlo + n * 64; u64* du = (u64*)(dhi + n * 64);
    khost::Fld SF(khost::scalar_field_id(curve));
    khost::fe uu; memcpy(&uu, u, 32);
    uu = SF.from_mont(uu);                                 // canonical integer for the double-and-add
    hipStream_t s = C.stream;
    KH_HIP(hipMemcpyAsync(dlo, g_lo, n * 64, hipMemcpyHostToDevice, s));
    KH_HIP(hipMemcpyAsync(dhi, g_hi, n * 64, hipMemcpyHostToDevice, s));
    KH_HIP(hipMemcpyAsync(du, &uu, 32, hipMemcpyHostToDevice, s));
    KH_HIP(hipStreamSynchronize(s));                       // uu is a stack buffer
    dim3 grid((unsigned)((n + 127) / 128));
    if (curve == KH_CURVE_VESTA) hipLaunchKernelGGL((k_fold_points<FqParams>), grid, dim3(128), 0, s, dlo, dhi, du, n, g_ipa_b.as<uint8_t>(), g_ipa_c.as<uint8_t>());
    else hipLaunchKernelGGL((k_fold_points<FpParams>), grid, dim3(128), 0, s, dlo, dhi, du, n, g_ipa_b.as<uint8_t>(), g_ipa_c.as<uint8_t>());
    KH_HIP(hipGetLastError());
    KH_HIP(hipMemcpyAsync(out_xy, g_ipa_b.p, n * 64, hipMemcpyDeviceToHost, s));
    KH_HIP(hipMemcpyAsync(out_inf, g_ipa_c.p, n, hipMemcpyDeviceToHost, s));
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}

int ipa_fold_points_endo(Context& C, int curve, const uint64_t* g_lo, const uint64_t* g_hi, const uint64_t chal[2], size_t n,
                         uint64_t* out_xy, uint8_t* out_inf) {
    int rc;
    if ((rc = g_ipa_a.reserve(n * 128 + 32))) return rc;
    if ((rc = g_ipa_b.reserve(n * 64))) return rc;
    if ((rc = g_ipa_c.reserve(n))) return rc;
    uint8_t* dlo = g_ipa_a.as<uint8_t>(); uint8_t* dhi = dlo + n * 64; u64* deq = (u64*)(dhi + n * 64);
    khost::fe eq; endo_coefficient(khost::base_field_id(curve), eq.l);
    hipStream_t s = C.stream;
    KH_HIP(hipMemcpyAsync(dlo, g_lo, n * 64, hipMemcpyHostToDevice, s));
    KH_HIP(hipMemcpyAsync(dhi, g_hi, n * 64, hipMemcpyHostToDevice, s));
    KH_HIP(hipMemcpyAsync(deq, &eq, 32, hipMemcpyHostToDevice, s));
    KH_HIP(hipStreamSynchronize(s));
    dim3 grid((unsigned)((n + 127) / 128));
    if (curve == KH_CURVE_VESTA) hipLaunchKernelGGL((k_fold_points_endo<FqParams>), grid, dim3(128), 0, s, dlo, dhi, chal[0], chal[1], deq, n, g_ipa_b.as<uint8_t>(), g_ipa_c.as<uint8_t>());
    else hipLaunchKernelGGL((k_fold_points_endo<FpParams>), grid, dim3(128), 0, s, dlo, dhi, chal[0], chal[1], deq, n, g_ipa_b.as<uint8_t>(), g_ipa_c.as<uint8_t>());
    KH_HIP(hipGetLastError());
    KH_HIP(hipMemcpyAsync(out_xy, g_ipa_b.p, n * 64, hipMemcpyDeviceToHost, s));
    KH_HIP(hipMemcpyAsync(out_inf, g_ipa_c.p, n, hipMemcpyDeviceToHost, s));
    KH_HIP(hipStreamSynchronize(s));
    return KH_OK;
}

}  // namespace kh
