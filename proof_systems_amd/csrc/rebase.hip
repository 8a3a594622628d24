// rebase.hip -- the opening rounds' LATE rounds over a materialised folded basis.
//
// Reference: SRS::open folds the basis after every round, g_{j}[i] = g_{j-1}[i] + [u_j] g_{j-1}[i + N_j] (poly-commitment/src/ipa.rs:985-1003,
// combine.rs:292-340).  csrc/ipa.hip never folds: round j's L / R are MSMs over the ORIGINAL window tables with the scalars
// a[t mod N_j] * coef_j[t div N_j] (coef_j = the tensor of (1, u_k)), i.e. every round does full-size MSM work -- 2 x n x 16 table additions and a
// reduction over 2^15 buckets, ~280 us of dependent kernels -- however short the vectors have become.  This file materialises the folded basis ONCE,
// in the background, after j0 rounds:
//
//      g'[i] = sum_{q < 2^j0} coef_j0[q] * G[q N + i],   i < N = n / 2^j0
//
// with its own window tables (narrower windows: the bucket reduction of a tail round shrinks with them), and the rounds switch over to it as soon as it
// is ready: from then on a round is an MSM over N + 2 points.  The group elements L_j, R_j, sg are the same (g_j[i] = sum_{q'} coef_rel[q'] g'[q' N_j + i],
// coef_rel = the first 2^(j - j0) entries of coef_j), so the proof bytes are.
//
// The materialisation is a batch of N MSMs of 2^j0 terms that SHARE their scalars: with the original tables T[w][t] = 2^(16 w) G_t it is, for every
// output, a sum of 16 x 2^j0 table points with signed 16-bit digits d.  |d| = 256 h + l splits it into two bucket sets per output (255 "lo" buckets by l,
// 128 "hi" buckets by h); the term lists per bucket are the SAME for every output, so the accumulation runs with lanes = consecutive outputs
// (uniform control flow, 4 KB coalesced table reads per wave):
//
//   k_rb_plan     one block: coef -> signed digits -> the term list of each of the 383 buckets
//   k_rb_acc      wave (bucket, 64 outputs): B[bucket][i] = sum of its terms' table points (mixed additions)
//   k_rb_reduce1  quad (chunk of 16 buckets, output): running sums -> (sum_j (j + 1) B_j, sum_j B_j) per chunk
//   k_rb_reduce2  quad per (output, set): chunks -> sum l B_l resp. sum h B_h (lane-cooperative additions, coop.cuh); lo + 2^8 hi
//   k_rb_tables   quad per point (the N outputs, then H and U): 2^(c w) P for every window w by lane-cooperative doublings, as XYZZ
//   k_rb_normalize thread per point: every level to affine with one inversion (an identity output abandons the rebase)
//
// Everything here is throughput work on a side stream; the round in flight keeps the latency path.
#include "common.hpp"
#include "curve.cuh"
#include "coop.cuh"
#include "host_ec.hpp"
#include "msm.hpp"

namespace kh {

static constexpr u32 RB_LO = 255, RB_HI = 128, RB_BUCKETS = RB_LO + RB_HI;      // lo buckets l = 1..255, hi buckets h = 1..128
static constexpr u32 RB_CHUNK = 16, RB_LO_CHUNKS = 16, RB_HI_CHUNKS = 8, RB_CHUNKS = RB_LO_CHUNKS + RB_HI_CHUNKS;

// ---- plan: digits of the Q shared scalars, term lists per bucket.  list entry = q << 5 | w << 1 | negative.
template <class SF>
__global__ void __launch_bounds__(1024)
k_rb_plan(const u64* __restrict__ coef, u32 Q, u32* __restrict__ off /* RB_BUCKETS + 1 */, u32* __restrict__ list, int32_t* __restrict__ dig /* 16 x Q scratch */) {
    __shared__ u32 cnt[RB_BUCKETS + 1], cur[RB_BUCKETS + 1];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i <= RB_BUCKETS; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (u32 q = tid; q < Q; q += 1024) {
        const Fe<SF> s = from_mont<SF>(Fe<SF>::load(coef + 4 * (size_t)q));
        u32 l[8];
#pragma unroll
        for (int t = 0; t < 8; t++) l[t] = s.v[t];
        u32 carry = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {                    // signed digits in (-2^15, 2^15]: sum_w d_w 2^(16 w) = s (s < 2^255: no carry out of the top digit)
            const u32 v = ((l[w >> 1] >> (16 * (w & 1))) & 0xffffu) + carry;
            int32_t d;
            if (v > 0x8000u) { d = (int32_t)v - 0x10000; carry = 1; } else { d = (int32_t)v; carry = 0; }
            dig[(size_t)w * Q + q] = d;
            const u32 m = (u32)(d < 0 ? -d : d), lo = m & 255u, hi = m >> 8;
            if (lo) atomicAdd(&cnt[lo - 1], 1u);
            if (hi) atomicAdd(&cnt[RB_LO + hi - 1], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) { u32 run = 0; for (u32 b = 0; b < RB_BUCKETS; b++) { const u32 c = cnt[b]; cur[b] = run; off[b] = run; run += c; } off[RB_BUCKETS] = run; }
    __syncthreads();
    for (u32 q = tid; q < Q; q += 1024) {
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int32_t d = dig[(size_t)w * Q + q];
            const u32 m = (u32)(d < 0 ? -d : d), lo = m & 255u, hi = m >> 8;
            const u32 enc = (q << 5) | ((u32)w << 1) | (d < 0 ? 1u : 0u);
            if (lo) list[atomicAdd(&cur[lo - 1], 1u)] = enc;
            if (hi) list[atomicAdd(&cur[RB_LO + hi - 1], 1u)] = enc;
        }
    }
}

// ---- accumulate: wave (bucket b, outputs i0 .. i0 + 63)
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_acc(const u32* __restrict__ off, const u32* __restrict__ list, const uint8_t* __restrict__ tables, size_t stride, u32 N, uint8_t* __restrict__ B) {
    const u32 b = blockIdx.x, i = blockIdx.y * 64 + threadIdx.x;
    const u32 e0 = off[b], e1 = off[b + 1];
    Xyzz<BF> acc = Xyzz<BF>::identity();
    for (u32 e = e0; e < e1; e++) {
        const u32 enc = list[e], q = enc >> 5, w = (enc >> 1) & 15u;
        const Aff<BF> p = Aff<BF>::load(tables + ((size_t)w * stride + (size_t)q * N + i) * 64);
        acc = madd<BF>(acc, p, (enc & 1u) != 0);
    }
    acc.store(B + ((size_t)b * N + i) * 128);
}

// ---- reduce, level 1: a QUAD per (chunk k, output i) -- 16 outputs per wave -- over the chunk's buckets j = 0 .. 15 (weights j + 1 inside the chunk):
//      part[k][i] = (A = sum_j (j + 1) B_j, S = sum_j B_j); the chunk's share of the weighted sum is A + 16 k' S (k' = the chunk's index inside its set).
//      (One lane per output measured 419 us for N = 4096: 32 dependent full additions at ~13 us each; the lane-cooperative addition is 5 product rounds.)
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_reduce1(const uint8_t* __restrict__ B, u32 N, uint8_t* __restrict__ part) {
    const u32 k = blockIdx.x, i = blockIdx.y * 16 + (threadIdx.x >> 2);
    const bool hi = k >= RB_LO_CHUNKS;
    const u32 first = hi ? RB_LO + (k - RB_LO_CHUNKS) * RB_CHUNK : k * RB_CHUNK;
    const u32 count = (!hi && k == RB_LO_CHUNKS - 1) ? RB_CHUNK - 1 : RB_CHUNK;          // the lo set has 255 buckets: its last chunk holds 15
    Fe<BF> run = quad_identity<BF>(), acc = quad_identity<BF>();
    for (int j = (int)count - 1; j >= 0; j--) {
        run = quad_add<BF>(run, quad_load<BF>(B + ((size_t)(first + j) * N + i) * 128));
        acc = quad_add<BF>(acc, run);
    }
    quad_store<BF>(part + (((size_t)k * 2) * N + i) * 128, acc);
    quad_store<BF>(part + (((size_t)k * 2 + 1) * N + i) * 128, run);
}

// ---- reduce, level 2: a quad per (output, set); a wave holds 8 outputs x 2 sets.  set value = sum_k A_k + 16 sum_k k S_k; out = lo + 2^8 hi
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_reduce2(const uint8_t* __restrict__ part, u32 N, uint8_t* __restrict__ out) {
    const u32 quad = threadIdx.x >> 2, set = quad & 1u;
    const u32 i = blockIdx.x * 8 + (quad >> 1);
    const bool live = i < N;
    const u32 ii = live ? i : N - 1;
    const u32 k0 = set ? RB_LO_CHUNKS : 0u, nk = set ? RB_HI_CHUNKS : RB_LO_CHUNKS;
    Fe<BF> sumA = quad_identity<BF>(), run = quad_identity<BF>(), wsum = quad_identity<BF>();
    for (int k = (int)RB_LO_CHUNKS - 1; k >= 0; k--) {      // uniform control flow for the wave: the hi quads (8 chunks) add identities in the upper half
        const bool has = (u32)k < nk;
        const u32 kk = has ? (u32)k : 0u;
        Fe<BF> A = quad_load<BF>(part + (((size_t)(k0 + kk) * 2) * N + ii) * 128), Sk = quad_load<BF>(part + (((size_t)(k0 + kk) * 2 + 1) * N + ii) * 128);
        if (!has) { A = quad_identity<BF>(); Sk = quad_identity<BF>(); }
        sumA = quad_add<BF>(sumA, A);
        if (k >= 1) {
            run = quad_add<BF>(run, Sk);
            wsum = quad_add<BF>(wsum, run);                  // after the loop: sum_k k S_k
        }
    }
    for (int t = 0; t < 4; t++) wsum = quad_dbl<BF>(wsum);   // x 16
    Fe<BF> r = quad_add<BF>(sumA, wsum);
    // hi quads: x 2^8, then the lo quad (one quad below) takes it
    Fe<BF> h = r;
    for (int t = 0; t < 8; t++) h = quad_dbl<BF>(h);
    const Fe<BF> up = quad_shfl_down<BF>(h, 1);
    r = quad_add<BF>(r, up);                                 // meaningful in the lo quads
    if (live && set == 0) quad_store<BF>(out + (size_t)i * 128, r);
}

// ---- the window tables of the new basis: a quad per point (N outputs as XYZZ from k_rb_reduce2, then H and U, affine, in the two extra slots).
//      scratch[w][i] = 2^(c w) P_i as XYZZ for w < W: c lane-cooperative doublings per level (4 product rounds each: the chain of ~250 doublings is
//      pure latency, one lane per point measured 1.29 ms for it).  k_rb_normalize then brings every level to affine with ONE inversion per point.
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_tables(const uint8_t* __restrict__ outs, const uint8_t* __restrict__ hu_affine, u32 N, u32 npts, int c, int W, uint8_t* __restrict__ scratch) {
    const u32 i = blockIdx.x * 16 + (threadIdx.x >> 2), role = threadIdx.x & 3u;
    const u32 ii = i < npts ? i : npts - 1;
    Fe<BF> P;
    if (ii < N) P = quad_load<BF>(outs + (size_t)ii * 128);
    else {                                                 // H, U: affine (x | y), ZZ = ZZZ = 1
        P = role < 2 ? Fe<BF>::load(hu_affine + (size_t)(ii - N) * 64 + 32 * role) : Fe<BF>::one();
    }
    for (int w = 0; w < W; w++) {
        if (w) for (int t = 0; t < c; t++) P = quad_dbl<BF>(P);
        if (i < npts) quad_store<BF>(scratch + ((size_t)w * npts + i) * 128, P);
    }
}
// thread per point: tables[w][i] = affine(scratch[w][i]) for every level through the running products of the ZZZ's and one inversion (Montgomery's trick
// over the levels, as msm.hip's k_precompute does); a point at infinity has no affine form: *fail is set and the caller keeps the original basis
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_normalize(const uint8_t* __restrict__ scratch, u32 npts, int W, uint8_t* __restrict__ tables, u32* __restrict__ fail) {
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    if (i >= npts) return;
    Fe<BF> prod = Fe<BF>::one();
    for (int w = 0; w < W; w++) {                          // prefix products, parked in the x slot of the (not yet written) table entry
        const Fe<BF> zzz = Fe<BF>::load(scratch + ((size_t)w * npts + i) * 128 + 96);
        prod.store(tables + ((size_t)w * npts + i) * 64);
        prod = mul<BF>(prod, zzz);
    }
    if (prod.is_zero()) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    Fe<BF> iv = inv<BF>(prod);                             // 1 / (zzz_0 ... zzz_{W-1})
    for (int w = W - 1; w >= 0; w--) {
        const Xyzz<BF> Q = Xyzz<BF>::load(scratch + ((size_t)w * npts + i) * 128);
        const Fe<BF> izzz = mul<BF>(iv, Fe<BF>::load(tables + ((size_t)w * npts + i) * 64));       // times the product of the levels below
        iv = mul<BF>(iv, Q.zzz);
        const Fe<BF> izz = sqr<BF>(mul<BF>(izzz, Q.zz));     // (ZZ / ZZZ)^2 = 1 / ZZ
        mul<BF>(Q.x, izz).store(tables + ((size_t)w * npts + i) * 64);
        mul<BF>(Q.y, izzz).store(tables + ((size_t)w * npts + i) * 64 + 32);
    }
}

size_t rebase_bucket_bytes(size_t N) { return (size_t)RB_BUCKETS * N * 128; }
size_t rebase_part_bytes(size_t N) { return (size_t)RB_CHUNKS * 2 * N * 128 + N * 128; }       // chunk pairs, then the N outputs (XYZZ)
size_t rebase_list_bytes(size_t Q) { return ((size_t)RB_BUCKETS + 1 + 2 * 16 * Q + 16 * Q) * 4; }

template <class BF, class SF>
static int rebase_t(hipStream_t s, const u64* coef, size_t Q, const void* tables, size_t stride, size_t N, uint8_t* B, uint8_t* part, u32* lists, hipEvent_t after_plan) {
    u32* off = lists; u32* list = off + RB_BUCKETS + 1; int32_t* dig = (int32_t*)(list + 2 * 16 * Q);
    uint8_t* outs = part + (size_t)RB_CHUNKS * 2 * N * 128;
    hipLaunchKernelGGL((k_rb_plan<SF>), dim3(1), dim3(1024), 0, s, coef, (u32)Q, off, list, dig);
    if (after_plan) KH_HIP(hipEventRecord(after_plan, s));   // `coef` is the caller's again once this event has passed
    hipLaunchKernelGGL((k_rb_acc<BF>), dim3(RB_BUCKETS, (unsigned)(N / 64)), dim3(64), 0, s, off, list, (const uint8_t*)tables, stride, (u32)N, B);
    hipLaunchKernelGGL((k_rb_reduce1<BF>), dim3(RB_CHUNKS, (unsigned)(N / 16)), dim3(64), 0, s, B, (u32)N, part);
    hipLaunchKernelGGL((k_rb_reduce2<BF>), dim3((unsigned)((N + 7) / 8)), dim3(64), 0, s, part, (u32)N, outs);
    KH_HIP(hipGetLastError());
    return KH_OK;
}
// g'[i] = sum_{q < Q} coef[q] * G[q N + i] for i < N (N a multiple of 64, Q < 2^22) from the c = 16 window tables `tables` (stride points per table), left as
// XYZZ records behind the chunk sums in `part` (rebase_outputs); everything queued on `s`.  B / part / lists: workspaces of rebase_*_bytes.  after_plan
// (nullable): recorded behind the only kernel that reads `coef`.
int rebase_points(hipStream_t s, int curve, const uint64_t* coef, size_t Q, const void* tables, size_t stride, size_t N, void* B, void* part, void* lists,
                  hipEvent_t after_plan) {
    KH_REQUIRE(N >= 64 && N % 64 == 0 && Q >= 1 && Q < ((size_t)1 << 22), "rebase_points: N = %zu, Q = %zu out of range", N, Q);
    if (curve == KH_CURVE_VESTA) return rebase_t<FqParams, FpParams>(s, coef, Q, tables, stride, N, (uint8_t*)B, (uint8_t*)part, (u32*)lists, after_plan);
    return rebase_t<FpParams, FqParams>(s, coef, Q, tables, stride, N, (uint8_t*)B, (uint8_t*)part, (u32*)lists, after_plan);
}
const void* rebase_outputs(const void* part, size_t N) { return (const uint8_t*)part + (size_t)RB_CHUNKS * 2 * N * 128; }
// The window tables (width c) of the N materialised points and of the `extra` affine points `extra_affine` (device memory, 64 bytes each: H and U) behind
// them: tables[w][i] = 2^(c w) P_i, affine, W x (N + extra) entries; scratch = W x (N + extra) x 128 bytes.  *fail != 0 afterwards: some point was the
// identity (no affine form).
int rebase_tables(hipStream_t s, int curve, const void* part, size_t N, const void* extra_affine, size_t extra, int c, void* scratch, void* tables, uint32_t* fail) {
    const int W = (256 + c - 1) / c;
    const u32 npts = (u32)(N + extra);
    const uint8_t* outs = (const uint8_t*)rebase_outputs(part, N);
    if (curve == KH_CURVE_VESTA) {
        hipLaunchKernelGGL((k_rb_tables<FqParams>), dim3((npts + 15) / 16), dim3(64), 0, s, outs, (const uint8_t*)extra_affine, (u32)N, npts, c, W, (uint8_t*)scratch);
        hipLaunchKernelGGL((k_rb_normalize<FqParams>), dim3((npts + 63) / 64), dim3(64), 0, s, (const uint8_t*)scratch, npts, W, (uint8_t*)tables, fail);
    } else {
        hipLaunchKernelGGL((k_rb_tables<FpParams>), dim3((npts + 15) / 16), dim3(64), 0, s, outs, (const uint8_t*)extra_affine, (u32)N, npts, c, W, (uint8_t*)scratch);
        hipLaunchKernelGGL((k_rb_normalize<FpParams>), dim3((npts + 63) / 64), dim3(64), 0, s, (const uint8_t*)scratch, npts, W, (uint8_t*)tables, fail);
    }
    KH_HIP(hipGetLastError());
    return KH_OK;
}

}  // namespace kh
