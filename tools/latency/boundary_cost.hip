// What does a stream of SHORT kernels cost a LONG kernel that runs beside it on another stream?  (Four provers share the chip at 175-200 proofs/s whatever is
// varied on the host side; every proof is ~270 kernel launches.)  A = a gather-heavy kernel over a 64 MB table (the shape of the MSM accumulation: random
// 64-byte reads that live in the L2 / the Infinity Cache), ~0.5 ms alone.  B = chains of one-block kernels (each ends with the release a kernel boundary
// carries, each start with the acquire) on 0 / 1 / 3 other streams while A runs; variant "rmw": the short kernels also dirty 1 MB each.
// Build: hipcc --offload-arch=gfx950 -O2 -o boundary_cost boundary_cost.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, size_t mask, unsigned iters, unsigned* __restrict__ out) {
    unsigned x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (unsigned i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint4 v = table[(x >> 4) & mask];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        x ^= acc & 0xffu;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_short(unsigned* data, size_t words, unsigned tag) {
    for (size_t k = threadIdx.x; k < words; k += blockDim.x) data[k] += tag;
}
static double med(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const size_t TBL = 64u << 20;
    uint4* table; CK(hipMalloc(&table, TBL)); CK(hipMemset(table, 1, TBL));
    unsigned* out; CK(hipMalloc(&out, 64));
    unsigned* scratch; CK(hipMalloc(&scratch, 16u << 20)); CK(hipMemset(scratch, 0, 16u << 20));
    hipStream_t sa, sb[3];
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    for (auto& s : sb) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = 256 * 8, iters = 70;
    for (int variant = 0; variant < 2; variant++) {
        const size_t words = variant ? (1u << 20) / 4 : 0;
        for (int nb : {0, 1, 3}) {
            for (int per_stream : {200, 600}) {
                if (nb == 0 && per_stream != 200) continue;
                std::vector<float> t;
                for (int r = 0; r < 12; r++) {
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, sa));
                    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, sa, table, TBL / 16 - 1, iters, out);
                    CK(hipEventRecord(e1, sa));
                    for (int i = 0; i < per_stream; i++)
                        for (int b = 0; b < nb; b++) hipLaunchKernelGGL(k_short, dim3(1), dim3(256), 0, sb[b], scratch + (size_t)b * (1u << 20), words, (unsigned)i);
                    CK(hipDeviceSynchronize());
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) t.push_back(ms * 1e3f);
                }
                printf("%-28s long kernel beside %d stream(s) x %3d short kernels: %7.1f us\n", variant ? "short kernels dirty 1 MB" : "short kernels empty", nb, per_stream, med(t));
            }
        }
    }
    return 0;
}
