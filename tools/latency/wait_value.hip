// hipStreamWaitValue32 as a gate WITHOUT a resident wave: the command processor polls a signal word; the host opens it with a plain store.
// Build: hipcc --offload-arch=gfx950 -O2 -o wait_value wait_value.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef std::chrono::steady_clock clk;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
__global__ void k_after(volatile unsigned* ack, unsigned tag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((unsigned*)ack, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    volatile unsigned* f; CK(hipHostMalloc((void**)&f, 4096, hipHostMallocDefault));
    for (int i = 0; i < 1024; i++) f[i] = 0;
    void* sig = nullptr;
    CK(hipExtMallocWithFlags(&sig, 8, hipMallocSignalMemory));
    printf("signal memory at %p\n", sig); fflush(stdout);
    hipPointerAttribute_t at; if (hipPointerGetAttributes(&at, sig) == hipSuccess) printf("  type %d host %p device %p\n", (int)at.type, at.hostPointer, at.devicePointer);
    fflush(stdout);
    *(volatile unsigned long long*)sig = 0;                  // is it host-writable?
    printf("host store to the signal word: ok\n"); fflush(stdout);
    std::vector<double> g;
    for (int r = 1; r <= 200; r++) {
        CK(hipStreamSynchronize(s));
        CK(hipStreamWaitValue32(s, sig, (unsigned)r, hipStreamWaitValueEq, 0xffffffffu));
        hipLaunchKernelGGL(k_after, dim3(64), dim3(256), 0, s, f + 64, (unsigned)r);
        const auto w0 = clk::now();
        while (us(w0, clk::now()) < 50.0) _mm_pause();
        if (f[64] == (unsigned)r) { printf("the kernel ran BEFORE the value was written: the wait does not hold\n"); return 1; }
        const auto t0 = clk::now();
        __atomic_store_n((volatile unsigned*)sig, (unsigned)r, __ATOMIC_RELEASE);
        while (f[64] != (unsigned)r) { if (us(t0, clk::now()) > 2e6) { printf("no wake-up within 2 s of the host store (rep %d)\n", r); return 2; } _mm_pause(); }
        const auto t1 = clk::now();
        if (r > 20) g.push_back(us(t0, t1));
    }
    printf("E  host store to the signal word -> first store of the kernel behind hipStreamWaitValue32   %6.1f us (min %.1f max %.1f)\n", med(g), *std::min_element(g.begin(), g.end()), *std::max_element(g.begin(), g.end()));
    return 0;
}
