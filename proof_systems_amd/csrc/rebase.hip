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
// output, a sum of 16 x 2^j0 table points with signed 16-bit digits d.  |d| = 1024 h + 32 m + l splits it into THREE bucket sets per output (31 buckets by l,
// 31 by m, 32 by h: 94 in all); the term lists per bucket are the SAME for every output, so the accumulation runs with lanes = consecutive outputs
// (uniform control flow, 4 KB coalesced table reads per wave).  (Until late in round 6 the split was 256 h + l: two sets, 383 buckets -- a third fewer
// additions in k_rb_acc, but four times the buckets to reduce per output: the two reduction kernels were 1.09 of the 2.15 ms the materialisation takes
// on its side stream, and the rounds switch over when it is done.)
//
//   k_rb_plan     one block: coef -> signed digits -> the term list of each of the 94 buckets
//   k_rb_acc      wave (bucket, 64 outputs): B[bucket][i] = sum of its terms' table points (mixed additions)
//   k_rb_reduce1  quad (chunk of 8 buckets, output): running sums -> (sum_j (j + 1) B_j, sum_j B_j) per chunk
//   k_rb_reduce2  quad per (output, set): chunks -> sum l B_l, sum m B_m, sum h B_h (lane-cooperative additions, coop.cuh); lo + 2^5 (mid + 2^5 hi)
//   k_rb_tables   quad per point (the N outputs, then H and U): 2^(c w) P for every window w by lane-cooperative doublings, as XYZZ
//   k_rb_normalize thread per point: every level to affine with one inversion (an identity output abandons the rebase)
//
// Everything here is throughput work on a side stream; the round in flight keeps the latency path.
#include <algorithm>
#include "common.hpp"
#include "curve.cuh"
#include "coop.cuh"
#include "field29.cuh"
#include "host_ec.hpp"
#include "msm.hpp"

namespace kh {

// |d| <= 2^15 = 1024 h + 32 m + l: buckets l = 1..31 (set 0), m = 1..31 (set 1), h = 1..32 (set 2); set s starts at bucket 31 s; chunks of 8 buckets, 4 per set
static constexpr u32 RB_BITS = 5, RB_SETS = 3, RB_SET_SIZE = 31, RB_BUCKETS = 2 * RB_SET_SIZE + 32;
static constexpr u32 RB_CHUNK = 8, RB_SET_CHUNKS = 4, RB_CHUNKS = RB_SETS * RB_SET_CHUNKS;
__device__ __forceinline__ void rb_split(u32 m, u32 b[3]) { b[0] = m & 31u; b[1] = (m >> 5) & 31u; b[2] = m >> 10; }

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
            u32 b[3]; rb_split((u32)(d < 0 ? -d : d), b);
#pragma unroll
            for (u32 t = 0; t < RB_SETS; t++) if (b[t]) atomicAdd(&cnt[RB_SET_SIZE * t + b[t] - 1], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) { u32 run = 0; for (u32 b = 0; b < RB_BUCKETS; b++) { const u32 c = cnt[b]; cur[b] = run; off[b] = run; run += c; } off[RB_BUCKETS] = run; }
    __syncthreads();
    for (u32 q = tid; q < Q; q += 1024) {
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int32_t d = dig[(size_t)w * Q + q];
            u32 b[3]; rb_split((u32)(d < 0 ? -d : d), b);
            const u32 enc = (q << 5) | ((u32)w << 1) | (d < 0 ? 1u : 0u);
#pragma unroll
            for (u32 t = 0; t < RB_SETS; t++) if (b[t]) list[atomicAdd(&cur[RB_SET_SIZE * t + b[t] - 1], 1u)] = enc;
        }
    }
}

// ---- accumulate: wave (bucket b, outputs i0 .. i0 + 63)
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_acc(const u32* __restrict__ off, const u32* __restrict__ list, const uint8_t* __restrict__ tables, size_t stride, u32 N, uint8_t* __restrict__ B) {
    const u32 b = blockIdx.x, i = blockIdx.y * 64 + threadIdx.x;
    const u32 e0 = off[b], e1 = off[b + 1];
    // the lazy 29-bit arithmetic of the MSM's accumulation (field29.cuh: ~1.5 x the plain mixed addition's rate; with three bucket sets this kernel is the
    // materialisation's largest after the table chain); a lane whose lazy sum cannot exclude an exceptional case (equal or opposite points: degenerate bases)
    // redoes its bucket with the exact formulas
    bool lazy_done = false;
    if (e1 > e0) {
        u32 enc = list[e0];
        Aff<BF> p = Aff<BF>::load(tables + ((size_t)((enc >> 1) & 15u) * stride + (size_t)(enc >> 5) * N + i) * 64);
        if (enc & 1u) p.y = neg<BF>(p.y);
        typedef typename C29<BF>::T K29;
        Acc29<BF> a29;
        a29.x = to29<BF>(p.x); a29.y = to29<BF>(p.y);
#pragma unroll
        for (int t = 0; t < 9; t++) { a29.zz.v[t] = K29::one(t); a29.zzz.v[t] = K29::one(t); }
        bool ok = true;
        for (u32 e = e0 + 1; e < e1; e++) {
            enc = list[e];
            p = Aff<BF>::load(tables + ((size_t)((enc >> 1) & 15u) * stride + (size_t)(enc >> 5) * N + i) * 64);
            if (enc & 1u) p.y = neg<BF>(p.y);
            ok = madd29<BF>(a29, pack29<BF, 5>(p.x), pack29<BF, 5>(p.y));
            if (!ok) break;
        }
        if (ok) { xyzz_from29<BF>(a29).store(B + ((size_t)b * N + i) * 128); lazy_done = true; }
    }
    if (lazy_done) return;
    Xyzz<BF> acc = Xyzz<BF>::identity();
    for (u32 e = e0; e < e1; e++) {
        const u32 enc = list[e], q = enc >> 5, w = (enc >> 1) & 15u;
        const Aff<BF> p = Aff<BF>::load(tables + ((size_t)w * stride + (size_t)q * N + i) * 64);
        acc = madd<BF>(acc, p, (enc & 1u) != 0);
    }
    acc.store(B + ((size_t)b * N + i) * 128);
}

// ---- reduce, level 1: a QUAD per (chunk k, output i) -- 16 outputs per wave -- over the chunk's buckets j = 0 .. 7 (weights j + 1 inside the chunk):
//      part[k][i] = (A = sum_j (j + 1) B_j, S = sum_j B_j); the chunk's share of the weighted sum is A + 8 k' S (k' = the chunk's index inside its set).
//      (One lane per output measured 419 us for N = 4096: 32 dependent full additions at ~13 us each; the lane-cooperative addition is 5 product rounds.)
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_reduce1(const uint8_t* __restrict__ B, u32 N, uint8_t* __restrict__ part) {
    const u32 k = blockIdx.x, i = blockIdx.y * 16 + (threadIdx.x >> 2);
    const u32 set = k / RB_SET_CHUNKS, kk = k - set * RB_SET_CHUNKS;
    const u32 first = RB_SET_SIZE * set + kk * RB_CHUNK;
    const u32 count = (set < 2 && kk == RB_SET_CHUNKS - 1) ? RB_CHUNK - 1 : RB_CHUNK;    // the sets of 31 buckets: the last chunk holds 7
    Fe<BF> run = quad_identity<BF>(), acc = quad_identity<BF>();
    for (int j = (int)count - 1; j >= 0; j--) {
        run = quad_add<BF>(run, quad_load<BF>(B + ((size_t)(first + j) * N + i) * 128));
        acc = quad_add<BF>(acc, run);
    }
    quad_store<BF>(part + (((size_t)k * 2) * N + i) * 128, acc);
    quad_store<BF>(part + (((size_t)k * 2 + 1) * N + i) * 128, run);
}

// ---- reduce, level 2: a quad per (output, set); a wave holds 4 outputs x (3 sets + an idle quad).  set value = sum_k A_k + 8 sum_k k S_k;
//      out = lo + 2^5 (mid + 2^5 hi): the hi quad's value travels down through the mid quad to the lo quad
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_reduce2(const uint8_t* __restrict__ part, u32 N, uint8_t* __restrict__ out) {
    const u32 quad = threadIdx.x >> 2, set = quad & 3u;
    const u32 i = blockIdx.x * 4 + (quad >> 2);
    const bool live = i < N;
    const u32 ii = live ? i : N - 1;
    const bool real = set < RB_SETS;
    const u32 k0 = (real ? set : 0u) * RB_SET_CHUNKS;
    Fe<BF> sumA = quad_identity<BF>(), run = quad_identity<BF>(), wsum = quad_identity<BF>();
    for (int k = (int)RB_SET_CHUNKS - 1; k >= 0; k--) {     // uniform control flow for the wave: the idle quads add identities
        Fe<BF> A = quad_load<BF>(part + (((size_t)(k0 + k) * 2) * N + ii) * 128), Sk = quad_load<BF>(part + (((size_t)(k0 + k) * 2 + 1) * N + ii) * 128);
        if (!real) { A = quad_identity<BF>(); Sk = quad_identity<BF>(); }
        sumA = quad_add<BF>(sumA, A);
        if (k >= 1) {
            run = quad_add<BF>(run, Sk);
            wsum = quad_add<BF>(wsum, run);                  // after the loop: sum_k k S_k
        }
    }
    for (int t = 0; t < 3; t++) wsum = quad_dbl<BF>(wsum);   // x 8
    Fe<BF> r = quad_add<BF>(sumA, wsum);
    // Horner over the sets: (hi 2^5 + mid) 2^5 + lo; after each step the quad one below holds the partial result (meaningful in the mid, then the lo quads)
    for (int step = 0; step < 2; step++) {
        Fe<BF> h = r;
        for (u32 t = 0; t < RB_BITS; t++) h = quad_dbl<BF>(h);
        const Fe<BF> up = quad_shfl_down<BF>(h, 1);
        const Fe<BF> sum = quad_add<BF>(r, up);
        // step 0: the mid quads take hi 2^5 + mid; step 1: the lo quads take (that) 2^5 + lo.  Other quads keep their own value for the next step.
        if (set == 1u - (u32)step) r = sum;
    }
    if (live && set == 0) quad_store<BF>(out + (size_t)i * 128, r);
}

// ---- the window tables of the new basis: a quad per point (N outputs as XYZZ from k_rb_reduce2, then H and U, affine, in the two extra slots).
//      scratch[w][i] = 2^(c w) P_i as XYZZ for w < W: c lane-cooperative doublings per level (4 product rounds each: the chain of ~250 doublings is
//      pure latency, one lane per point measured 1.29 ms for it).  k_rb_normalize then brings every level to affine with ONE inversion per point.
__device__ __forceinline__ void rb_setprio(int p) { if (p == 3) __builtin_amdgcn_s_setprio(3); else if (p == 2) __builtin_amdgcn_s_setprio(2); else if (p == 1) __builtin_amdgcn_s_setprio(1); }
struct RbBeta { u64 l[4]; };
struct RbExtra { u64 xy[2][8]; };                       // up to two extra affine points (H, U), in the kernel's ARGUMENTS: a copy from the caller's pageable memory
                                                          // queued behind the materialisation made the HOST wait for it (hipMemcpyAsync from pageable memory returns when
                                                          // the stream has reached the copy): ~1 ms of the round that launched the rebase, in every opening, until found
template <class BF>
__global__ void __launch_bounds__(64)
k_rb_tables(const uint8_t* __restrict__ outs, RbExtra hu, u32 N, u32 npts, int c, int W, uint8_t* __restrict__ scratch, int prio) {
    // ~130 waves walking a chain of ~250 dependent doublings: pure latency, and the long pole of the materialisation.  Its wave priority (KH_IPA_REBASE_PRIO)
    // moves WHERE the time goes, not how much: at 3 (above the rounds' kernels) the tables are ready four rounds earlier -- 8.1 instead of 4.0 rounds per
    // opening run over the folded basis -- and the three rounds that share their SIMDs with it take 430 / 390 / 830-870 us instead of 296; at 0 (default) the
    // chain gets the issue slots the rounds leave and they stay at 296.  Opening 4.97-5.03 ms either way (2, 1: the same); so did a partition of the compute
    // units between the chain and the rounds (CU-masked streams: the masked rounds cost 40 us each, and one test run hung).
    rb_setprio(prio);
    const u32 i = blockIdx.x * 16 + (threadIdx.x >> 2), role = threadIdx.x & 3u;
    const u32 ii = i < npts ? i : npts - 1;
    Fe<BF> P;
    if (ii < N) P = quad_load<BF>(outs + (size_t)ii * 128);
    else {                                                 // H, U: affine (x | y), ZZ = ZZZ = 1
        P = role < 2 ? Fe<BF>::load(&hu.xy[ii - N][4 * role]) : Fe<BF>::one();
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
k_rb_normalize(const uint8_t* __restrict__ scratch, u32 npts, int W, uint8_t* __restrict__ tables, u32* __restrict__ fail, int prio, RbBeta beta, int glv) {
    rb_setprio(prio);                                      // (33 waves, one inversion deep: as k_rb_tables)
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
        const Fe<BF> ax = mul<BF>(Q.x, izz), ay = mul<BF>(Q.y, izzz);
        ax.store(tables + ((size_t)w * npts + i) * 64);
        ay.store(tables + ((size_t)w * npts + i) * 64 + 32);
        if (glv) {                                         // phi(x, y) = (beta x, y) = [lambda] (x, y): the level the second half-scalar's digits index
            mul<BF>(ax, Fe<BF>::load(beta.l)).store(tables + ((size_t)(w + W) * npts + i) * 64);
            ay.store(tables + ((size_t)(w + W) * npts + i) * 64 + 32);
        }
    }
}

size_t rebase_bucket_bytes(size_t N) { return (size_t)RB_BUCKETS * N * 128; }
size_t rebase_part_bytes(size_t N) { return (size_t)RB_CHUNKS * 2 * N * 128 + N * 128; }       // chunk pairs, then the N outputs (XYZZ)
size_t rebase_list_bytes(size_t Q) { return ((size_t)RB_BUCKETS + 1 + RB_SETS * 16 * Q + 16 * Q) * 4; }

template <class BF, class SF>
static int rebase_t(hipStream_t s, const u64* coef, size_t Q, const void* tables, size_t stride, size_t N, uint8_t* B, uint8_t* part, u32* lists, hipEvent_t after_plan) {
    u32* off = lists; u32* list = off + RB_BUCKETS + 1; int32_t* dig = (int32_t*)(list + RB_SETS * 16 * Q);
    uint8_t* outs = part + (size_t)RB_CHUNKS * 2 * N * 128;
    hipLaunchKernelGGL((k_rb_plan<SF>), dim3(1), dim3(1024), 0, s, coef, (u32)Q, off, list, dig);
    if (after_plan) KH_HIP(hipEventRecord(after_plan, s));   // `coef` is the caller's again once this event has passed
    // The two kernels with thousands of waves are held to a few waves per CU (dynamic LDS they do not use, as k_acc_wide29 is held): resident all at once they
    // took the wave slots and registers of every CU for ~0.3 ms, and the round in flight -- whose kernels outrank them in issue priority but have to be PLACED
    // first -- stalled for that long (a 5 us k_digits took 73, the round 1.2-1.36 ms instead of 0.35).  KH_IPA_REBASE_WAVES_PER_CU (experiment, default 0 = no limit: 2 / 4 / 8 waves per CU measured 5.28 / 5.20 / 5.06 ms per opening against 4.97-5.01 -- the stall was the host's, see RbExtra).
    static const unsigned rb_waves = getenv("KH_IPA_REBASE_WAVES_PER_CU") ? (unsigned)atoi(getenv("KH_IPA_REBASE_WAVES_PER_CU")) : 0u;
    const size_t hold = (rb_waves == 0 || rb_waves >= 32) ? 0 : std::min<size_t>(65536, (((size_t)160 << 10) / (rb_waves + 1) + 1024) & ~(size_t)1023);
    hipLaunchKernelGGL((k_rb_acc<BF>), dim3(RB_BUCKETS, (unsigned)(N / 64)), dim3(64), hold, s, off, list, (const uint8_t*)tables, stride, (u32)N, B);
    hipLaunchKernelGGL((k_rb_reduce1<BF>), dim3(RB_CHUNKS, (unsigned)(N / 16)), dim3(64), hold, s, B, (u32)N, part);
    hipLaunchKernelGGL((k_rb_reduce2<BF>), dim3((unsigned)((N + 3) / 4)), dim3(64), 0, s, part, (u32)N, outs);
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
// The window tables (width c) of the N materialised points and of the `extra` (<= 2) affine points `extra_affine_host` (HOST memory, 64 bytes each: H and U) behind
// them: tables[w][i] = 2^(c w) P_i, affine, W x (N + extra) entries; scratch = W x (N + extra) x 128 bytes.  *fail != 0 afterwards: some point was the
// identity (no affine form).
int rebase_tables(hipStream_t s, int curve, const void* part, size_t N, const void* extra_affine_host, size_t extra, int c, void* scratch, void* tables, uint32_t* fail,
                  const uint64_t* glv_beta) {
    KH_REQUIRE(extra <= 2 && (extra == 0 || extra_affine_host), "rebase_tables: at most two extra points");
    // glv_beta: W = the levels built by doubling (the lower 128 bits: HALF the chain); the kernel that normalises them writes phi of every level W levels further up
    const int glv = glv_beta ? 1 : 0;
    const int W = glv ? (128 + c - 1) / c : (256 + c - 1) / c;
    RbBeta beta; memset(&beta, 0, sizeof beta);
    if (glv) memcpy(beta.l, glv_beta, 32);
    const u32 npts = (u32)(N + extra);
    static const int rb_prio = getenv("KH_IPA_REBASE_PRIO") ? atoi(getenv("KH_IPA_REBASE_PRIO")) : 0;
    RbExtra hu; memset(&hu, 0, sizeof hu);
    if (extra) memcpy(&hu, extra_affine_host, 64 * extra);
    const uint8_t* outs = (const uint8_t*)rebase_outputs(part, N);
    if (curve == KH_CURVE_VESTA) {
        hipLaunchKernelGGL((k_rb_tables<FqParams>), dim3((npts + 15) / 16), dim3(64), 0, s, outs, hu, (u32)N, npts, c, W, (uint8_t*)scratch, rb_prio);
        hipLaunchKernelGGL((k_rb_normalize<FqParams>), dim3((npts + 63) / 64), dim3(64), 0, s, (const uint8_t*)scratch, npts, W, (uint8_t*)tables, fail, rb_prio, beta, glv);
    } else {
        hipLaunchKernelGGL((k_rb_tables<FpParams>), dim3((npts + 15) / 16), dim3(64), 0, s, outs, hu, (u32)N, npts, c, W, (uint8_t*)scratch, rb_prio);
        hipLaunchKernelGGL((k_rb_normalize<FpParams>), dim3((npts + 63) / 64), dim3(64), 0, s, (const uint8_t*)scratch, npts, W, (uint8_t*)tables, fail, rb_prio, beta, glv);
    }
    KH_HIP(hipGetLastError());
    return KH_OK;
}

}  // namespace kh
