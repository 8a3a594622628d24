// Host Poseidon permutation rate through the C ABI sponge (kh_sponge_*): the transcript is the strictly sequential part of a proof.
// Build: clang++ -O3 -std=c++17 tools/poseidon_bench.cpp proof_systems_amd/csrc/host_sponge.cpp -o /tmp/poseidon_bench
#include <chrono>
#include <cstdio>
#include "../include/kimchi_hip.h"
namespace kh { void set_error(const char*, ...) {} }
int main() {
    for (int kind = 0; kind < 2; kind++) {
        kh_sponge_t* s; kh_sponge_new(kind, 0, &s);
        uint64_t x[4] = {1, 2, 3, 0}, out[4];
        const int N = 40000;                                    // absorbs; one permutation per two absorbed elements
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) kh_sponge_absorb(s, x, 1);
        auto t1 = std::chrono::steady_clock::now();
        kh_sponge_squeeze_field(s, out);
        printf("sponge kind %d: %.2f us per permutation (%016llx)\n", kind, std::chrono::duration<double, std::micro>(t1 - t0).count() / (N / 2), (unsigned long long)out[0]);
        kh_sponge_free(s);
    }
}
