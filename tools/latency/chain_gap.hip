// Idle time between DEPENDENT kernels of one stream when the host has queued them ahead -- measured by the kernels themselves (every kernel stores the
// wall clock at its first and after its last instruction), not by a profiler: are the 6-12 us the rocprofv3 trace shows between the kernels of an opening
// round (plain launches) real, and does a replayed graph close them?  Build: hipcc --offload-arch=gfx950 -O2 -o chain_gap chain_gap.hip
//   plain   N launches behind a 300 us spin kernel (the host is far ahead when the chain starts)
//   graph   the same N launches captured once, replayed behind the spin kernel
// for three kernel shapes: one block; 2048 blocks x 256 threads touching 64 MB (a release that has dirty lines to write back); 256 blocks with 64 KB of dynamic LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(unsigned long long ticks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4); }
}
// every block of link i stores its own first and last clock (the host takes the earliest start and the latest end: 2 x 2048 atomics on one address took 44 us
// to drain behind an otherwise empty kernel and showed up as a "gap" in the first version of this file)
__global__ void k_link(unsigned long long* stamps, unsigned i, unsigned* data, size_t words) {
    extern __shared__ unsigned lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned long long* mine = stamps + 2 * ((size_t)i * gridDim.x + blockIdx.x);
    if (words) {
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < words; k += stride) data[k] += i;
    }
    if (threadIdx.x == 0) lds[0] = i;
    __syncthreads();
    if (threadIdx.x == 0) { mine[0] = t0; mine[1] = wall_clock64(); }
}

struct Shape { const char* name; unsigned grid, block; size_t lds; size_t words; };

int main() {
    const int N = 12, REPS = 40;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t MAXG = 16384;
    unsigned long long* stamps; CK(hipMalloc(&stamps, 2 * N * MAXG * sizeof(unsigned long long)));
    std::vector<unsigned long long> raw(2 * N * MAXG);
    unsigned* data; CK(hipMalloc(&data, 64u << 20)); CK(hipMemset(data, 0, 64u << 20));
    std::vector<unsigned long long> h(2 * N), init(2 * N);
    for (int i = 0; i < N; i++) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    CK(hipFuncSetAttribute((const void*)k_link, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const Shape shapes[] = {{"1 block", 1, 64, 4, 0}, {"2048 x 256, 64 MB read-modify-write", 2048, 256, 4, (64u << 20) / 4}, {"256 x 256, 64 KB LDS", 256, 256, 65536, 0},
                            {"256 x 256, 4 B LDS", 256, 256, 4, 0}, {"256 x 256, 16 KB LDS", 256, 256, 16384, 0}, {"256 x 256, 32 KB LDS", 256, 256, 32768, 0},
                            {"1 x 256, 64 KB LDS", 1, 256, 65536, 0}, {"32 x 256, 64 KB LDS", 32, 256, 65536, 0}, {"2048 x 256, 4 B LDS", 2048, 256, 4, 0}, {"16384 x 256, 4 B LDS", 16384, 256, 4, 0},
                            {"256 x 1024, 4 B LDS", 256, 1024, 4, 0}, {"2048 x 256, 4 MB read-modify-write", 2048, 256, 4, (4u << 20) / 4}};
    for (const Shape& sh : shapes) {
        for (int mode = 0; mode < 2; mode++) {
            hipGraphExec_t ge = nullptr;
            if (mode == 1) {
                hipGraph_t g; CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
                for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_link, dim3(sh.grid), dim3(sh.block), sh.lds, s, stamps, (unsigned)i, data, sh.words);
                CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
            }
            std::vector<double> gaps, durs, total;
            for (int r = 0; r < REPS + 3; r++) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 30000ull);      // 300 us at 100 MHz
                if (mode == 0) for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_link, dim3(sh.grid), dim3(sh.block), sh.lds, s, stamps, (unsigned)i, data, sh.words);
                else CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(raw.data(), stamps, 2 * N * (size_t)sh.grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                h = init;
                for (int i = 0; i < N; i++) for (unsigned b = 0; b < sh.grid; b++) {
                    const unsigned long long* m = &raw[2 * ((size_t)i * sh.grid + b)];
                    h[2 * i] = std::min(h[2 * i], m[0]); h[2 * i + 1] = std::max(h[2 * i + 1], m[1]);
                }
                if (r < 3) continue;
                double g = 0, d = 0;
                for (int i = 1; i < N; i++) g += (double)((long long)(h[2 * i] - h[2 * i - 1])) / 100.0;
                for (int i = 0; i < N; i++) d += (double)(h[2 * i + 1] - h[2 * i]) / 100.0;
                gaps.push_back(g / (N - 1)); durs.push_back(d / N); total.push_back((double)(h[2 * N - 1] - h[0]) / 100.0);
            }
            std::sort(gaps.begin(), gaps.end()); std::sort(durs.begin(), durs.end()); std::sort(total.begin(), total.end());
            printf("%-38s %-5s  gap between links: median %5.2f us (min %5.2f, max %5.2f)   link %6.2f us   chain of %d: %7.1f us\n", sh.name, mode ? "graph" : "plain",
                   gaps[gaps.size() / 2], gaps.front(), gaps.back(), durs[durs.size() / 2], N, total[total.size() / 2]);
            if (ge) CK(hipGraphExecDestroy(ge));
        }
    }
    return 0;
}
