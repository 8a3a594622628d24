// Host field inversion (csrc/host_ec.hpp: batched division steps) against its definition a^(p-2), both Pasta fields: edge values and a
// seeded random sweep; prints the time per inversion of both.  Built and run by tests/test_host_inv.py (CPU, no GPU, no library).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../../proof_systems_amd/csrc/host_ec.hpp"
using namespace khost;
static u64 s = 0x9e3779b97f4a7c15ull;
static u64 rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200000;
    for (int id = 0; id < 2; id++) {
        Fld F(id);
        auto check = [&](const fe& a, const char* what) {
            const fe x = F.inv(a), y = F.inv_fermat(a);
            if (!eq(x, y)) { printf("MISMATCH field %d (%s): %016llx %016llx %016llx %016llx\n", id, what, (unsigned long long)a.l[3], (unsigned long long)a.l[2], (unsigned long long)a.l[1], (unsigned long long)a.l[0]); exit(1); }
            if (!is_zero(a) && !eq(F.mul(x, a), F.f.one)) { printf("not an inverse (field %d, %s)\n", id, what); exit(1); }
        };
        fe z = {{0, 0, 0, 0}}; check(z, "zero");
        check(F.f.one, "one"); check(F.neg(F.f.one), "minus one");
        fe pm1; sub_n(pm1, F.f.p, fe{{1, 0, 0, 0}}); check(pm1, "limbs p - 1");
        for (u64 k = 1; k < 70; k++) { fe a = {{k, 0, 0, 0}}; check(a, "small"); check(F.to_mont(a), "small (mont)"); fe b = {{0, 0, 0, 0}}; b.l[k / 18] = 1ull << ((k * 7) % 62); check(b, "one bit"); }
        for (int i = 0; i < N; i++) {
            fe a = {{rnd(), rnd(), rnd(), rnd() >> 2}};
            if (geq(a, F.f.p)) sub_n(a, a, F.f.p);
            if (i % 7 == 0) { a.l[3] = 0; a.l[2] = 0; }                 // short operands
            if (i % 11 == 0) { a.l[0] = 0; a.l[1] &= ~0xffffull; }      // many trailing zero bits
            check(a, "random");
        }
        fe a = {{rnd(), rnd(), rnd(), rnd() >> 3}}, acc = a;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20000; i++) acc = F.add(F.inv(acc), a);
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20000; i++) acc = F.add(F.inv_fermat(acc), a);
        auto t2 = std::chrono::steady_clock::now();
        printf("field %d: %d values ok; inv %.0f ns, a^(p-2) %.0f ns (%llx)\n", id, N, std::chrono::duration<double, std::nano>(t1 - t0).count() / 20000,
               std::chrono::duration<double, std::nano>(t2 - t1).count() / 20000, (unsigned long long)acc.l[0]);
    }
    printf("HOST_INV_OK\n");
    return 0;
}
