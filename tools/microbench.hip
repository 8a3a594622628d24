// Integer-ALU micro-benchmark for gfx950: issue rates of the instructions a
// 256-bit Montgomery multiplication is built from, plus candidate mont-mul bodies.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define ITERS 2048
template <int OP>
__global__ void k_rate(u32* out, u32 seed) {
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a0 = tid + seed, a1 = tid * 3 + seed, a2 = tid * 5 + 1, a3 = tid * 7 + 2, a4 = tid ^ seed, a5 = tid + 11, a6 = tid + 13, a7 = tid + 17;
    u32 x = tid * 2654435761u + 1, y = seed * 40503u + 3;
    double d0 = tid, d1 = tid + 1, d2 = tid + 2, d3 = tid + 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7, dx = 1.0000001, dy = 0.5;
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) {  // v_mad_u64_u32, 8 independent chains
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
        } else if (OP == 1) {  // v_mul_lo_u32
            u32 *p = (u32*)&a0;
            asm volatile(
                "v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\t"
                "v_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8"
                : "+v"(((u32*)&a0)[0]), "+v"(((u32*)&a1)[0]), "+v"(((u32*)&a2)[0]), "+v"(((u32*)&a3)[0]),
                  "+v"(((u32*)&a4)[0]), "+v"(((u32*)&a5)[0]), "+v"(((u32*)&a6)[0]), "+v"(((u32*)&a7)[0]) : "v"(x));
            (void)p;
        } else if (OP == 2) {  // v_mul_hi_u32
            asm volatile(
                "v_mul_hi_u32 %0, %0, %8\n\tv_mul_hi_u32 %1, %1, %8\n\tv_mul_hi_u32 %2, %2, %8\n\tv_mul_hi_u32 %3, %3, %8\n\t"
                "v_mul_hi_u32 %4, %4, %8\n\tv_mul_hi_u32 %5, %5, %8\n\tv_mul_hi_u32 %6, %6, %8\n\tv_mul_hi_u32 %7, %7, %8"
                : "+v"(((u32*)&a0)[0]), "+v"(((u32*)&a1)[0]), "+v"(((u32*)&a2)[0]), "+v"(((u32*)&a3)[0]),
                  "+v"(((u32*)&a4)[0]), "+v"(((u32*)&a5)[0]), "+v"(((u32*)&a6)[0]), "+v"(((u32*)&a7)[0]) : "v"(x));
        } else if (OP == 3) {  // v_addc_co_u32 chain (full-rate reference)
            asm volatile(
                "v_add_co_u32 %0, vcc, %0, %8\n\tv_addc_co_u32 %1, vcc, %1, %8, vcc\n\tv_addc_co_u32 %2, vcc, %2, %8, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
                "v_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_addc_co_u32 %5, vcc, %5, %8, vcc\n\tv_addc_co_u32 %6, vcc, %6, %8, vcc\n\tv_addc_co_u32 %7, vcc, %7, %8, vcc"
                : "+v"(((u32*)&a0)[0]), "+v"(((u32*)&a1)[0]), "+v"(((u32*)&a2)[0]), "+v"(((u32*)&a3)[0]),
                  "+v"(((u32*)&a4)[0]), "+v"(((u32*)&a5)[0]), "+v"(((u32*)&a6)[0]), "+v"(((u32*)&a7)[0]) : "v"(x) : "vcc");
        } else if (OP == 4) {  // v_mad_u32_u24
            asm volatile(
                "v_mad_u32_u24 %0, %0, %8, %9\n\tv_mad_u32_u24 %1, %1, %8, %9\n\tv_mad_u32_u24 %2, %2, %8, %9\n\tv_mad_u32_u24 %3, %3, %8, %9\n\t"
                "v_mad_u32_u24 %4, %4, %8, %9\n\tv_mad_u32_u24 %5, %5, %8, %9\n\tv_mad_u32_u24 %6, %6, %8, %9\n\tv_mad_u32_u24 %7, %7, %8, %9"
                : "+v"(((u32*)&a0)[0]), "+v"(((u32*)&a1)[0]), "+v"(((u32*)&a2)[0]), "+v"(((u32*)&a3)[0]),
                  "+v"(((u32*)&a4)[0]), "+v"(((u32*)&a5)[0]), "+v"(((u32*)&a6)[0]), "+v"(((u32*)&a7)[0]) : "v"(x), "v"(y));
        } else if (OP == 5) {  // v_fma_f64
            asm volatile(
                "v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx), "v"(dy));
        } else if (OP == 6) {  // v_mad_u64_u32 + v_addc pair (the MAC primitive), 4 chains
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
                "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
                "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(((u32*)&a4)[0]), "+v"(((u32*)&a5)[0]), "+v"(((u32*)&a6)[0]), "+v"(((u32*)&a7)[0]) : "v"(x), "v"(y) : "vcc");
        } else if (OP == 7) {  // single dependent chain of MAC pairs (latency view)
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                : "+v"(a0), "+v"(((u32*)&a4)[0]) : "v"(x), "v"(y) : "vcc");
        }
    }
    u64 s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    out[tid] = (u32)s ^ (u32)(s >> 32) ^ (u32)ds;
}

template <int OP>
int run_rate(const char* name, int ops_per_iter, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves: one per SIMD
    u32* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double lane_ops = (double)blocks * 256 * ITERS * ops_per_iter;
    double per_s = lane_ops / (ms * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    double wave_instr = lane_ops / 64.0;
    double cyc = (ms * 1e-3 * 2.4e9) * 1024.0 / wave_instr;
    printf("%-28s waves/SIMD=%d  %8.3f ms  %8.2f Tlane-op/s  %6.2f cyc/wave-instr/SIMD (at 2.4GHz)\n", name, waves_per_simd, ms, per_s / 1e12, cyc);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs=%d, clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    for (int w : {1, 2, 4, 8}) {
        run_rate<0>("v_mad_u64_u32 x8", 8, w);
        run_rate<6>("mad_u64_u32+addc x4", 8, w);
        run_rate<7>("mad+addc dependent chain", 8, w);
        run_rate<1>("v_mul_lo_u32 x8", 8, w);
        run_rate<2>("v_mul_hi_u32 x8", 8, w);
        run_rate<3>("v_addc_co_u32 x8", 8, w);
        run_rate<4>("v_mad_u32_u24 x8", 8, w);
        run_rate<5>("v_fma_f64 x8", 8, w);
    }
    return 0;
}
