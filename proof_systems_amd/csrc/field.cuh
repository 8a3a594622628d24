// field.cuh -- Pasta Fp / Fq arithmetic for gfx950 (CDNA4), device side.
//
// Element = 256-bit integer in Montgomery form, R = 2^256, fully reduced to
// [0, p): bit-identical to ark-ff's Fp256<MontBackend<_, 4>> in-memory value
// (curves/src/pasta/fields/fp.rs:8-12, fq.rs:8-12 of the reference), so host
// buffers of 4 x u64 limbs are used as they are.
//
// Multiplication is product-scanning (column-wise) Montgomery on eight 32-bit
// limbs, emitted as ONE hand-scheduled asm block (tools/gen_field_asm.py ->
// field_mulasm.inc); from_mont (the reduction half alone) uses the per-column
// blocks of field_cols.inc (tools/gen_field_cols.py).  Each limb product is ONE v_mad_u64_u32 whose carry-out (VCC) is
// collected by ONE v_addc_co_u32 into a third accumulator word: 96-bit column
// accumulator, no 64-bit adds, no compare-for-carry.  Both Pasta primes are
//     p = 2^254 + t * 2^32 + 1,  t < 2^96
// i.e. 32-bit words [1, p1, p2, p3, 0, 0, 0, 2^30].  Hence
//   * -p^-1 mod 2^32 = 0xffffffff, so the Montgomery quotient digit is just
//     m_k = -acc_k mod 2^32 (no multiply),
//   * m*p needs 3 real multiplies per digit (p1, p2, p3) plus m*2^30,
// giving 64 + 8*4 = 96 MADs per multiplication instead of the generic 136.
//
// The column blocks in field_cols.inc are generated (see tools/gen_field_cols.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kh {

typedef uint32_t u32;
typedef uint64_t u64;

#include "field_cols.inc"
#include "field_mulasm.inc"

struct FpParams {   // scalar field of Vesta, base field of Pallas
    static constexpr u32 P1 = 0x992d30edu, P2 = 0x094cf91bu, P3 = 0x224698fcu;
    __device__ static constexpr u32 one(int i) {   // R mod p
        constexpr u32 R[8] = {0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
        return R[i];
    }
    __device__ static constexpr u32 r2(int i) {    // R^2 mod p
        constexpr u32 R2[8] = {0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu};
        return R2[i];
    }
};
struct FqParams {   // base field of Vesta, scalar field of Pallas
    static constexpr u32 P1 = 0x8c46eb21u, P2 = 0x0994a8ddu, P3 = 0x224698fcu;
    __device__ static constexpr u32 one(int i) {
        constexpr u32 R[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
        return R[i];
    }
    __device__ static constexpr u32 r2(int i) {
        constexpr u32 R2[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
        return R2[i];
    }
};
static constexpr u32 P0W = 1u, P7W = 0x40000000u;

template <class F>
struct Fe {
    u32 v[8];

    __device__ __forceinline__ static Fe zero() { Fe r; _Pragma("unroll") for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
    __device__ __forceinline__ static Fe one() { Fe r; _Pragma("unroll") for (int i = 0; i < 8; i++) r.v[i] = F::one(i); return r; }
    __device__ __forceinline__ static Fe r2() { Fe r; _Pragma("unroll") for (int i = 0; i < 8; i++) r.v[i] = F::r2(i); return r; }
    __device__ __forceinline__ static u32 pw(int i) {
        return i == 0 ? P0W : i == 1 ? F::P1 : i == 2 ? F::P2 : i == 3 ? F::P3 : i == 7 ? P7W : 0u;
    }
    __device__ __forceinline__ bool is_zero() const {
        u32 o = 0; _Pragma("unroll") for (int i = 0; i < 8; i++) o |= v[i]; return o == 0;
    }
    __device__ __forceinline__ bool operator==(const Fe& b) const {
        u32 o = 0; _Pragma("unroll") for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i]; return o == 0;
    }
    // 32-byte global load/store as two 16-byte accesses
    __device__ __forceinline__ static Fe load(const void* p) {
        const uint4* q = (const uint4*)p; uint4 a = q[0], b = q[1];
        Fe r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        return r;
    }
    __device__ __forceinline__ void store(void* p) const {
        uint4* q = (uint4*)p;
        q[0] = make_uint4(v[0], v[1], v[2], v[3]); q[1] = make_uint4(v[4], v[5], v[6], v[7]);
    }
};

// r = t - p if t >= p else t   (t < 2p < 2^256)
template <class F>
__device__ __forceinline__ Fe<F> cond_sub_p(const u32 t[8]) {
    u32 s[8]; u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 d = (u64)t[i] - Fe<F>::pw(i) - br;
        s[i] = (u32)d; br = (u32)(d >> 63);
    }
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = br ? t[i] : s[i];
    return r;
}

#define KH_FE_BINOP_OPERANDS                                                                                              \
    : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])   \
    : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),              \
      "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]),              \
      "v"(p1), "v"(p2), "v"(p3)

// a + b mod p: carry chain, trial subtraction of p, select (24 instructions; the compiler's
// rendering of the same C expression was ~55: every madd does seven of these)
template <class F>
__device__ __forceinline__ Fe<F> add(const Fe<F>& a, const Fe<F>& b) {
    Fe<F> r;
    const u32 p1 = F::P1, p2 = F::P2, p3 = F::P3;
    asm(KH_FE_ADD_ASM KH_FE_BINOP_OPERANDS : KH_FE_ADD_CLOBBERS);
    return r;
}
// a - b mod p: borrow chain, masked add-back of p (22 instructions)
template <class F>
__device__ __forceinline__ Fe<F> sub(const Fe<F>& a, const Fe<F>& b) {
    Fe<F> r;
    const u32 p1 = F::P1, p2 = F::P2, p3 = F::P3;
    asm(KH_FE_SUB_ASM KH_FE_BINOP_OPERANDS : KH_FE_SUB_CLOBBERS);
    return r;
}
template <class F>
__device__ __forceinline__ Fe<F> neg(const Fe<F>& a) {
    Fe<F> r; u32 br = 0;
    u32 nz = a.is_zero() ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) { u64 d = (u64)Fe<F>::pw(i) - a.v[i] - br; r.v[i] = (u32)d & nz; br = (u32)(d >> 63); }
    return r;
}
template <class F>
__device__ __forceinline__ Fe<F> dbl(const Fe<F>& a) { return add<F>(a, a); }

// Montgomery product a*b*2^-256 mod p, fully reduced: one hand-scheduled asm block
// (tools/gen_field_asm.py: 254 instructions, 104 v_mad_u64_u32).
template <class F>
__device__ __forceinline__ Fe<F> mul(const Fe<F>& a, const Fe<F>& b) {
    Fe<F> r;
    const u32 p1 = F::P1, p2 = F::P2, p3 = F::P3;
    asm(KH_MONT_MUL_ASM
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]),
          "v"(p1), "v"(p2), "v"(p3)
        : KH_MONT_MUL_CLOBBERS);
    return r;
}
// a^2 * 2^-256 mod p: dedicated schedule, 36 limb products (tools/gen_field_asm.py gen_sqr)
template <class F>
__device__ __forceinline__ Fe<F> sqr(const Fe<F>& a) {
    Fe<F> r;
    const u32 p1 = F::P1, p2 = F::P2, p3 = F::P3;
    asm(KH_MONT_SQR_ASM
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(p1), "v"(p2), "v"(p3)
        : KH_MONT_SQR_CLOBBERS);
    return r;
}

// a * 2^-256 mod p  (Montgomery -> canonical integer): the reduction half only.
template <class F>
__device__ __forceinline__ Fe<F> from_mont(const Fe<F>& a) {
    u32 m[8]; u32 t[8];
    u64 lo; u32 hi = 0;
    const u32* A = a.v;
    const u32 P1 = F::P1, P2 = F::P2, P3 = F::P3, P7 = P7W;
#define KH_RED(k) { m[k] = 0u - (u32)lo; u32 c_ = ((u32)lo != 0u) ? 1u : 0u; lo = ((lo >> 32) | ((u64)hi << 32)) + c_; }
#define KH_OUT(k) { t[k - 8] = (u32)lo; lo = (lo >> 32) | ((u64)hi << 32); }
    lo = A[0]; KH_RED(0)
    lo += A[1]; col1(lo, hi, m[0],P1); KH_RED(1)
    lo += A[2]; col2(lo, hi, m[1],P1, m[0],P2); KH_RED(2)
    lo += A[3]; col3(lo, hi, m[2],P1, m[1],P2, m[0],P3); KH_RED(3)
    lo += A[4]; col3(lo, hi, m[3],P1, m[2],P2, m[1],P3); KH_RED(4)
    lo += A[5]; col3(lo, hi, m[4],P1, m[3],P2, m[2],P3); KH_RED(5)
    lo += A[6]; col3(lo, hi, m[5],P1, m[4],P2, m[3],P3); KH_RED(6)
    lo += A[7]; col4(lo, hi, m[6],P1, m[5],P2, m[4],P3, m[0],P7); KH_RED(7)
    col4(lo, hi, m[7],P1, m[6],P2, m[5],P3, m[1],P7); KH_OUT(8)
    col3(lo, hi, m[7],P2, m[6],P3, m[2],P7); KH_OUT(9)
    col2(lo, hi, m[7],P3, m[3],P7); KH_OUT(10)
    col1(lo, hi, m[4],P7); KH_OUT(11)
    col1(lo, hi, m[5],P7); KH_OUT(12)
    col1(lo, hi, m[6],P7); KH_OUT(13)
    col1(lo, hi, m[7],P7); KH_OUT(14)
    t[7] = (u32)lo;
#undef KH_RED
#undef KH_OUT
    return cond_sub_p<F>(t);
}
template <class F>
__device__ __forceinline__ Fe<F> to_mont(const Fe<F>& a) { return mul<F>(a, Fe<F>::r2()); }

// a^(p-2) = a^-1 (Fermat).  p - 2 has words [0xffffffff, P1 - 1, P2, P3, 0, 0, 0, 2^30].
template <class F>
__device__ __noinline__ Fe<F> inv(const Fe<F>& a) {
    const u32 e[8] = {0xffffffffu, F::P1 - 1u, F::P2, F::P3, 0u, 0u, 0u, P7W};
    Fe<F> acc = a;                       // top bit (254) consumed
    for (int i = 253; i >= 0; i--) {
        acc = sqr<F>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = mul<F>(acc, a);
    }
    return acc;
}

}  // namespace kh
