// VALU issue-rate micro-benchmark for gfx950: the instructions a 255-bit Montgomery multiplication is built from,
// CONTROL instructions (v_mov_b32, v_add_u32, v_and_b32, v_fma_f32, v_pk_fma_f32: what the chip's "full rate" is), and
// the complete product bodies of field.cuh (eight 32-bit limbs) and field29.cuh (nine 29-bit limbs).
// Every kernel runs >= 5 ms at 8 waves/SIMD (ITERS x 8 instructions per wave, 65536 iterations) so that launch
// overhead and clock ramp are < 1 %; rates are reported as cycles per wave-instruction per SIMD at the 2.4 GHz peak
// clock (the guide's "2 cycles per wave64 VALU op" vs a 4-cycle SIMD16-style issue is what the controls decide).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I proof_systems_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "field29.cuh"

using namespace kh;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define ITERS 65536
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

enum { OP_MOV, OP_ADD, OP_AND, OP_ADD3, OP_ADDC, OP_ALIGNBIT, OP_BFI, OP_LSHR64, OP_LSHLADD64, OP_MULLO, OP_MULHI, OP_MAD24, OP_MAD64,
       OP_MAD64_S, OP_MAD64_ADDC, OP_FMA32, OP_PKFMA32, OP_FMA64, OP_COUNT };

template <int OP>
__global__ void k_rate(u32* out, u32 seed, u32 sval) {
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a0 = tid + seed, a1 = tid * 3 + seed, a2 = tid * 5 + 1, a3 = tid * 7 + 2, a4 = tid ^ seed, a5 = tid + 11, a6 = tid + 13, a7 = tid + 17;
    u32 x = tid * 2654435761u + 1, y = seed * 40503u + 3;
    float f0 = tid, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7, fx = 1.0000001f, fy = 0.5f;
    double d0 = tid, d1 = tid + 1, d2 = tid + 2, d3 = tid + 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7, dx = 1.0000001, dy = 0.5;
#define LO(k) "+v"(((u32*)&a##k)[0])
    for (int i = 0; i < ITERS; i++) {
        if (OP == OP_MOV) {
            asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x));
        } else if (OP == OP_ADD) {
            asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x));
        } else if (OP == OP_AND) {
            asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x));
        } else if (OP == OP_ADD3) {
            asm volatile("v_add3_u32 %0, %0, %8, %9\n\tv_add3_u32 %1, %1, %8, %9\n\tv_add3_u32 %2, %2, %8, %9\n\tv_add3_u32 %3, %3, %8, %9\n\tv_add3_u32 %4, %4, %8, %9\n\tv_add3_u32 %5, %5, %8, %9\n\tv_add3_u32 %6, %6, %8, %9\n\tv_add3_u32 %7, %7, %8, %9"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x), "v"(y));
        } else if (OP == OP_ADDC) {
            asm volatile("v_add_co_u32 %0, vcc, %0, %8\n\tv_addc_co_u32 %1, vcc, %1, %8, vcc\n\tv_addc_co_u32 %2, vcc, %2, %8, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
                         "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_addc_co_u32 %5, vcc, %5, %8, vcc\n\tv_addc_co_u32 %6, vcc, %6, %8, vcc\n\tv_addc_co_u32 %7, vcc, %7, %8, vcc"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x) : "vcc");
        } else if (OP == OP_ALIGNBIT) {
            asm volatile("v_alignbit_b32 %0, %0, %8, 29\n\tv_alignbit_b32 %1, %1, %8, 29\n\tv_alignbit_b32 %2, %2, %8, 29\n\tv_alignbit_b32 %3, %3, %8, 29\n\tv_alignbit_b32 %4, %4, %8, 29\n\tv_alignbit_b32 %5, %5, %8, 29\n\tv_alignbit_b32 %6, %6, %8, 29\n\tv_alignbit_b32 %7, %7, %8, 29"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x));
        } else if (OP == OP_BFI) {
            asm volatile("v_bfi_b32 %0, %0, 0, %8\n\tv_bfi_b32 %1, %1, 0, %8\n\tv_bfi_b32 %2, %2, 0, %8\n\tv_bfi_b32 %3, %3, 0, %8\n\tv_bfi_b32 %4, %4, 0, %8\n\tv_bfi_b32 %5, %5, 0, %8\n\tv_bfi_b32 %6, %6, 0, %8\n\tv_bfi_b32 %7, %7, 0, %8"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "s"(sval));
        } else if (OP == OP_LSHR64) {
            asm volatile("v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\tv_lshrrev_b64 %3, 1, %3\n\tv_lshrrev_b64 %4, 1, %4\n\tv_lshrrev_b64 %5, 1, %5\n\tv_lshrrev_b64 %6, 1, %6\n\tv_lshrrev_b64 %7, 1, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == OP_LSHLADD64) {
            asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\tv_lshl_add_u64 %2, %2, 0, %8\n\tv_lshl_add_u64 %3, %3, 0, %8\n\tv_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\tv_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d0));
        } else if (OP == OP_MULLO) {
            asm volatile("v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\tv_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x));
        } else if (OP == OP_MULHI) {
            asm volatile("v_mul_hi_u32 %0, %0, %8\n\tv_mul_hi_u32 %1, %1, %8\n\tv_mul_hi_u32 %2, %2, %8\n\tv_mul_hi_u32 %3, %3, %8\n\tv_mul_hi_u32 %4, %4, %8\n\tv_mul_hi_u32 %5, %5, %8\n\tv_mul_hi_u32 %6, %6, %8\n\tv_mul_hi_u32 %7, %7, %8"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x));
        } else if (OP == OP_MAD24) {
            asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n\tv_mad_u32_u24 %1, %1, %8, %9\n\tv_mad_u32_u24 %2, %2, %8, %9\n\tv_mad_u32_u24 %3, %3, %8, %9\n\tv_mad_u32_u24 %4, %4, %8, %9\n\tv_mad_u32_u24 %5, %5, %8, %9\n\tv_mad_u32_u24 %6, %6, %8, %9\n\tv_mad_u32_u24 %7, %7, %8, %9"
                         : LO(0), LO(1), LO(2), LO(3), LO(4), LO(5), LO(6), LO(7) : "v"(x), "v"(y));
        } else if (OP == OP_MAD64) {
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
        } else if (OP == OP_MAD64_S) {            // one multiplicand in an SGPR (the reduction MADs of field29.cuh)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "s"(sval) : "vcc");
        } else if (OP == OP_MAD64_ADDC) {         // the MAC primitive of field.cuh: MAD + carry collection
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
                         "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), LO(4), LO(5), LO(6), LO(7) : "v"(x), "v"(y) : "vcc");
        } else if (OP == OP_FMA32) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fx), "v"(fy));
        } else if (OP == OP_PKFMA32) {            // two fp32 FMAs per lane per instruction
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n\tv_pk_fma_f32 %1, %1, %8, %8\n\tv_pk_fma_f32 %2, %2, %8, %8\n\tv_pk_fma_f32 %3, %3, %8, %8\n\tv_pk_fma_f32 %4, %4, %8, %8\n\tv_pk_fma_f32 %5, %5, %8, %8\n\tv_pk_fma_f32 %6, %6, %8, %8\n\tv_pk_fma_f32 %7, %7, %8, %8"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx));
        } else if (OP == OP_FMA64) {
            asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\tv_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx), "v"(dy));
        }
    }
    u64 s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    float fs = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    out[tid] = (u32)s ^ (u32)(s >> 32) ^ (u32)ds ^ (u32)fs;
}

// further 32-bit ops of the accumulation kernel's instruction mix: one kernel per mnemonic, "op dst, src, dst"
#define RATE2(NAME, ASM)                                                                                                  \
    __global__ void k_rate2_##NAME(u32* out, u32 seed) {                                                                  \
        u32 tid = blockIdx.x * blockDim.x + threadIdx.x;                                                                  \
        u32 a0 = tid + seed, a1 = tid * 3 + seed, a2 = tid * 5 + 1, a3 = tid * 7 + 2, a4 = tid ^ seed, a5 = tid + 11, a6 = tid + 13, a7 = tid + 17; \
        u32 x = (tid * 2654435761u + 1) & 15u, y = seed * 40503u + 3;                                                     \
        for (int i = 0; i < ITERS; i++)                                                                                   \
            asm volatile(ASM(0) "\n\t" ASM(1) "\n\t" ASM(2) "\n\t" ASM(3) "\n\t" ASM(4) "\n\t" ASM(5) "\n\t" ASM(6) "\n\t" ASM(7)   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc"); \
        out[tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                                 \
    }
#define A_SUB(k) "v_sub_u32 %" #k ", %" #k ", %8"
#define A_LSHR(k) "v_lshrrev_b32 %" #k ", %8, %" #k
#define A_LSHL(k) "v_lshlrev_b32 %" #k ", %8, %" #k
#define A_OR(k) "v_or_b32 %" #k ", %" #k ", %8"
#define A_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8"
#define A_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %9, vcc"
#define A_CMP(k) "v_cmp_le_u32 vcc, %" #k ", %9"
#define A_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 3, %9"
#define A_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9"
#define A_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %9"
#define A_SUBCO(k) "v_sub_co_u32 %" #k ", vcc, %" #k ", %8"
RATE2(sub, A_SUB) RATE2(lshr, A_LSHR) RATE2(lshl, A_LSHL) RATE2(or, A_OR) RATE2(xor, A_XOR) RATE2(cndmask, A_CNDMASK) RATE2(cmp, A_CMP)
RATE2(lshlor, A_LSHLOR) RATE2(andor, A_ANDOR) RATE2(lshladd, A_LSHLADD) RATE2(subco, A_SUBCO)

static int run_rate2(const char* name, void (*kern)(u32*, u32), int waves_per_simd) {
    int blocks = 256 * waves_per_simd;
    u32* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double wave_instr = (double)blocks * 4 * ITERS * 8;
    printf("%-30s waves/SIMD=%d  %8.3f ms  %7.2f G wave-instr/s  %5.2f cycles / wave-instr / SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           wave_instr / (ms * 1e-3) / 1e9, (ms * 1e-3 * 2.4e9) * 1024.0 / wave_instr);
    CHECK(hipFree(out));
    return 0;
}

static const char* OP_NAME[OP_COUNT] = {"v_mov_b32 (control)", "v_add_u32 (control)", "v_and_b32 (control)", "v_add3_u32", "v_addc_co_u32 chain", "v_alignbit_b32",
                                        "v_bfi_b32 (SGPR operand)", "v_lshrrev_b64", "v_lshl_add_u64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mad_u64_u32",
                                        "v_mad_u64_u32 (SGPR operand)", "v_mad_u64_u32 + v_addc pair", "v_fma_f32 (control)", "v_pk_fma_f32 (control)", "v_fma_f64"};

template <int OP>
int run_rate(int waves_per_simd) {
    int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves: one per SIMD on each of the 256 CUs
    u32* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u, 0x1fffffffu);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u, 0x1fffffffu);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double wave_instr = (double)blocks * 4 * ITERS * 8;
    double cyc = (ms * 1e-3 * 2.4e9) * 1024.0 / wave_instr;
    printf("%-30s waves/SIMD=%d  %8.3f ms  %7.2f G wave-instr/s  %5.2f cycles / wave-instr / SIMD (at 2.4 GHz)\n", OP_NAME[OP], waves_per_simd, ms,
           wave_instr / (ms * 1e-3) / 1e9, cyc);
    CHECK(hipFree(out));
    return 0;
}

// ------------------------------------------------------------------------------------------------ whole products
// dependent chains x <- x * y of the three product bodies; NPROD products per thread
#define NPROD 16384
template <int WHICH>
__global__ void k_prod(u32* out, u32 seed) {
    typedef FqParams F;
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (WHICH == 0) {
        Fe<F> x, y;
        for (int i = 0; i < 8; i++) { x.v[i] = tid * 2654435761u + i * seed; y.v[i] = tid + 40503u * i + seed; }
        x.v[7] &= 0x3fffffffu; y.v[7] &= 0x3fffffffu;
        for (int i = 0; i < NPROD; i++) x = mul<F>(x, y);
        u32 s = 0; for (int i = 0; i < 8; i++) s ^= x.v[i];
        out[tid] = s;
    } else {
        Fe29<F> x, y;
        for (int i = 0; i < 9; i++) { x.v[i] = (tid * 2654435761u + i * seed) & MASK29; y.v[i] = (tid + 40503u * i + seed) & MASK29; }
        x.v[8] &= 0x3fffffu; y.v[8] &= 0x3fffffu;
        typedef typename C29<F>::T K; const u32 p1 = K::P1, p2 = K::P2, p3 = K::P3, p4 = K::P4, c22 = 1u << 22, msk = MASK29, pairk = (1u << 29) + 1u;
        for (int i = 0; i < NPROD; i++) {
            Fe29<F> r;
            if (WHICH == 1) r = mul29<F>(x, y);
            else
                asm(KH29_MUL_ASM_B32
                    : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(r.v[8])
                    : "v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]), "v"(x.v[4]), "v"(x.v[5]), "v"(x.v[6]), "v"(x.v[7]), "v"(x.v[8]),
                      "v"(y.v[0]), "v"(y.v[1]), "v"(y.v[2]), "v"(y.v[3]), "v"(y.v[4]), "v"(y.v[5]), "v"(y.v[6]), "v"(y.v[7]), "v"(y.v[8]),
                      "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(c22), "s"(msk), "v"(pairk)
                    : "vcc", "v2", "v3");
            x = r;
        }
        u32 s = 0; for (int i = 0; i < 9; i++) s ^= x.v[i];
        out[tid] = s;
    }
}
template <int WHICH>
int run_prod(const char* name, int instr, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;
    u32* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_prod<WHICH>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_prod<WHICH>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double prods = (double)blocks * 256 * NPROD;
    double wave_instr = prods / 64 * instr;
    printf("%-44s waves/SIMD=%d  %8.3f ms  %7.2f G products/s  %6.1f cycles / product / SIMD  (%d instr: %5.2f cycles / instr)\n", name, waves_per_simd, ms,
           prods / (ms * 1e-3) / 1e9, (ms * 1e-3 * 2.4e9) * 1024.0 / (prods / 64), instr, (ms * 1e-3 * 2.4e9) * 1024.0 / wave_instr);
    CHECK(hipFree(out));
    return 0;
}

template <int OP> int sweep_rate() { for (int w : {1, 2, 4, 8}) if (run_rate<OP>(w)) return 1; return 0; }

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs=%d, clock=%d kHz; %d iterations x 8 instructions per wave\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, ITERS);
    if (sweep_rate<OP_MOV>() || sweep_rate<OP_ADD>() || sweep_rate<OP_AND>() || sweep_rate<OP_FMA32>() || sweep_rate<OP_PKFMA32>() || sweep_rate<OP_FMA64>() ||
        sweep_rate<OP_ADD3>() || sweep_rate<OP_ADDC>() || sweep_rate<OP_ALIGNBIT>() || sweep_rate<OP_BFI>() || sweep_rate<OP_LSHR64>() || sweep_rate<OP_LSHLADD64>() ||
        sweep_rate<OP_MULLO>() || sweep_rate<OP_MULHI>() || sweep_rate<OP_MAD24>() || sweep_rate<OP_MAD64>() || sweep_rate<OP_MAD64_S>() || sweep_rate<OP_MAD64_ADDC>()) return 1;
    struct { const char* n; void (*k)(u32*, u32); } more[] = {{"v_sub_u32", k_rate2_sub}, {"v_lshrrev_b32", k_rate2_lshr}, {"v_lshlrev_b32", k_rate2_lshl}, {"v_or_b32", k_rate2_or},
        {"v_xor_b32", k_rate2_xor}, {"v_cndmask_b32", k_rate2_cndmask}, {"v_cmp_le_u32 (-> vcc)", k_rate2_cmp}, {"v_lshl_or_b32", k_rate2_lshlor}, {"v_and_or_b32", k_rate2_andor},
        {"v_lshl_add_u32", k_rate2_lshladd}, {"v_sub_co_u32 (-> vcc)", k_rate2_subco}};
    for (auto& m : more) for (int w : {4, 8}) if (run_rate2(m.n, m.k, w)) return 1;
    for (int w : {1, 2, 4, 8}) {
        if (run_prod<0>("Montgomery product, 8 x 32-bit limbs (field.cuh)", 254, w)) return 1;
        if (run_prod<1>("Montgomery product, 9 x 29-bit limbs (field29)", 166, w)) return 1;
        if (run_prod<2>("  same with v_alignbit + v_lshrrev_b32 shifts", 182, w)) return 1;
    }
    return 0;
}
