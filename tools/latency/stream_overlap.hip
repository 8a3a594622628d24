// How many streams of one process run kernels at the same time?  T streams, each a chain of 40 one-block kernels that spin for 50 us: perfect overlap = 2 ms
// whatever T is; streams that share a hardware queue (GPU_MAX_HW_QUEUES, default 4) serialise.  Build: hipcc --offload-arch=gfx950 -O2 -o stream_overlap stream_overlap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_spin(unsigned long long ticks) {
    if (threadIdx.x == 0) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4); }
}
int main() {
    const int MAXT = 32, N = 40;
    std::vector<hipStream_t> s(MAXT);
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int T : {1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32}) {
        double best = 1e9;
        for (int r = 0; r < 4; r++) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) for (int t = 0; t < T; t++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[t], 5000ull);
            CK(hipDeviceSynchronize());
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("%2d streams x %d kernels of 50 us: %6.2f ms  (one stream alone: %.2f ms) -> %.1f streams' worth of overlap\n", T, N, best, N * 0.05, T * N * 0.05 / best);
    }
    return 0;
}
