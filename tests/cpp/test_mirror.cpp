// Exercises the C++ host mirror (include/kimchi_hip.hpp) the way the reference's own tests use
// the SRS trait: poly-commitment/tests/ipa_commitment.rs:26-52 (Lagrange commitments equal
// commit(interpolate(e_i))) restated on interpolate/evaluate + commit identities that need no
// external oracle, and the BlindersDontMatch error path (ipa.rs:611-613).
// Build: g++ -std=c++17 -Iinclude tests/cpp/test_mirror.cpp -Lproof_systems_amd -lkimchi_hip
#include <cstdio>
#include <cstring>
#include <random>

#include "kimchi_hip.hpp"

using namespace kimchi_hip;

static Fe rnd(std::mt19937_64& g) { Fe f{g(), g(), g(), g() & ((1ull << 61) - 1)}; return f; }   // < 2^253 < p: valid Montgomery limbs
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    check(kh_init(0));
    std::mt19937_64 gen(42);
    const size_t n = 128;
    SRS srs = SRS::create(Curve::Vesta, n);
    REQUIRE(srs.max_poly_size() == n && srs.size() == n);
    auto d1 = Radix2EvaluationDomain::create(Field::Fp, n);
    auto d8 = Radix2EvaluationDomain::create(Field::Fp, 8 * n);

    // interpolate(evaluate_over_domain(p)) == p, and d1 is every 8th point of d8 (kimchi/tests/test_domain.rs:25-71)
    DensePolynomial p; for (size_t i = 0; i < n; i++) p.coeffs.push_back(rnd(gen));
    Evaluations e1{p.evaluate_over_domain(d1), d1};
    Evaluations e8{p.evaluate_over_domain(d8), d8};
    for (size_t i = 0; i < n; i++) REQUIRE(e8.evals[8 * i] == e1.evals[i]);
    DensePolynomial back = e1.interpolate();
    REQUIRE(back.coeffs == p.coeffs);

    // chunking: 300 coefficients over an SRS of 128 -> 3 chunks, padded to 6 (tests/commitment.rs:348-386 shape)
    DensePolynomial big; for (size_t i = 0; i < 301; i++) big.coeffs.push_back(rnd(gen));
    PolyComm c6 = srs.commit_non_hiding(big, 6);
    REQUIRE(c6.chunks.size() == 6 && !c6.chunks[2].infinity && c6.chunks[3].infinity && c6.chunks[5].infinity);
    PolyComm c1 = srs.commit_non_hiding(big, 1);
    REQUIRE(c1.chunks.size() == 3);                      // never truncated
    for (int j = 0; j < 3; j++) REQUIRE(c1.chunks[j].x == c6.chunks[j].x && c1.chunks[j].y == c6.chunks[j].y);
    DensePolynomial zero; zero.coeffs.assign(10, Fe{});
    PolyComm cz = srs.commit_non_hiding(zero, 1);
    REQUIRE(cz.chunks.size() == 1 && cz.chunks[0].infinity);

    // linearity: commit(p) + 0*h == mask_custom(commit(p), [0])
    ScalarPolyComm zb; zb.chunks.assign(1, Fe{});
    BlindedCommitment m = srs.mask_custom(srs.commit_non_hiding(p, 1), zb);
    PolyComm cp = srs.commit_non_hiding(p, 1);
    REQUIRE(m.commitment.chunks[0].x == cp.chunks[0].x && m.commitment.chunks[0].y == cp.chunks[0].y);

    // BlindersDontMatch
    bool threw = false;
    try { ScalarPolyComm two; two.chunks.assign(2, Fe{}); srs.mask_custom(cp, two); } catch (const Error& e) { threw = e.code == KH_E_BLINDERS; }
    REQUIRE(threw);
    std::printf("MIRROR_OK\n");
    return 0;
}
