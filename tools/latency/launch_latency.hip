// Host <-> GPU hand-over latencies on one stream (what an opening round pays twice: the host learns that the round's last kernel has ended, and the next
// round's first kernel starts after the host has the challenge).  Build: hipcc --offload-arch=gfx950 -O2 -o launch_latency launch_latency.hip
//   A  launch on an idle stream -> first instruction (the kernel writes a pinned flag at its start)
//   B  kernel end -> host, through hipEventQuery polling vs through a pinned flag the kernel's last instruction writes
//   C  a kernel that is ALREADY queued and spins on a pinned flag the host sets ("gate"): host store -> kernel's acknowledgement
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

__global__ void k_work(volatile unsigned* start_flag, volatile unsigned* end_flag, unsigned tag, unsigned spin_ticks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __hip_atomic_store((unsigned*)start_flag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(2);
        __hip_atomic_store((unsigned*)end_flag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_gate(volatile unsigned* go, volatile unsigned* ack, unsigned tag, unsigned long long max_ticks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load((unsigned*)go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
            if (wall_clock64() - t0 > max_ticks) break;        // bounded: never hang the box
            __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store((unsigned*)ack, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_after(volatile unsigned* ack, unsigned tag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((unsigned*)ack, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    CK(hipSetDevice(0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    volatile unsigned* f; CK(hipHostMalloc((void**)&f, 4096, hipHostMallocDefault));
    for (int i = 0; i < 1024; i++) f[i] = 0;
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int reps = 300;
    std::vector<double> a_start, b_flag, b_event, c_gate, c_gate_next, l_call;
    for (int r = 1; r <= reps; r++) {
        CK(hipStreamSynchronize(s));
        const auto t0 = clk::now();
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, f, f + 16, (unsigned)r, 1000u);      // ~10 us at 100 MHz
        CK(hipEventRecord(ev, s));
        const auto t1 = clk::now();
        while (f[0] != (unsigned)r) _mm_pause();
        const auto t2 = clk::now();
        bool got_flag = false, got_ev = false; clk::time_point tf, te;
        while (!got_flag || !got_ev) {
            if (!got_flag && f[16] == (unsigned)r) { tf = clk::now(); got_flag = true; }
            if (!got_ev && hipEventQuery(ev) == hipSuccess) { te = clk::now(); got_ev = true; }
        }
        if (r > 20) { l_call.push_back(us(t0, t1)); a_start.push_back(us(t0, t2)); b_flag.push_back(us(t2, tf) - 10.0); b_event.push_back(us(t2, te) - 10.0); }
    }
    for (int r = 1; r <= reps; r++) {                       // B': only the event is polled (no flag reads in between)
        CK(hipStreamSynchronize(s));
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, f, f + 16, (unsigned)(r + 1000), 1000u);
        CK(hipEventRecord(ev, s));
        while (f[0] != (unsigned)(r + 1000)) _mm_pause();
        const auto t2 = clk::now();
        while (hipEventQuery(ev) != hipSuccess) _mm_pause();
        const auto te = clk::now();
        (void)hipGetLastError();
        if (r > 20) c_gate_next.push_back(us(t2, te) - 10.0);
    }
    printf("A  launch call (kernel + event record) returns after        %6.1f us\n", med(l_call));
    printf("A  launch call -> kernel's first store seen by the host     %6.1f us\n", med(a_start));
    printf("B  kernel's last store -> host, pinned flag                 %6.1f us   (after the ~10 us the kernel runs)\n", med(b_flag));
    printf("B  kernel end -> host, hipEventQuery polling (with flag)    %6.1f us\n", med(b_event));
    printf("B' kernel end -> host, hipEventQuery polling only           %6.1f us\n", med(c_gate_next));
    std::vector<double> g1, g2;
    for (int r = 1; r <= reps; r++) {
        CK(hipStreamSynchronize(s));
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s, f + 32, f + 48, (unsigned)r, 100ull * 200000ull);       // gives up after 0.2 s
        hipLaunchKernelGGL(k_after, dim3(64), dim3(256), 0, s, f + 64, (unsigned)r);
        const auto w0 = clk::now();
        while (us(w0, clk::now()) < 50.0) _mm_pause();      // the gate is resident and spinning by now
        const auto t0 = clk::now();
        f[32] = (unsigned)r;
        _mm_sfence();
        while (f[48] != (unsigned)r) _mm_pause();
        const auto t1 = clk::now();
        while (f[64] != (unsigned)r) _mm_pause();
        const auto t2 = clk::now();
        if (r > 20) { g1.push_back(us(t0, t1)); g2.push_back(us(t0, t2)); }
    }
    printf("C  host store -> spinning gate kernel's acknowledgement     %6.1f us\n", med(g1));
    printf("C  host store -> first store of the kernel queued behind it %6.1f us\n", med(g2));
    return 0;
}
