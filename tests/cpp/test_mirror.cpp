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
    // opening rounds with a = b = e_0 and zero blinders: a_hi = 0 so L_0 is the identity, R_0 = <a_lo, g_hi> + <a_lo, b_hi> U
    // = g[n/2] exactly; a and b stay e_0 through every fold, so a0 = b0 = 1
    {
        uint64_t one[4]; check(kh_domain_generator(int(Field::Fp), 0, one));      // omega_1 = 1 in Montgomery form
        Fe fone{one[0], one[1], one[2], one[3]};
        std::vector<Fe> a(1, fone), b(n, Fe{}); b[0] = fone;
        std::vector<uint64_t> g(8 * n); check(kh_srs_get_g(srs.raw(), 0, n, g.data()));
        Affine u_base = srs.blinding_commitment();
        OpeningRounds st(srs, a, b, u_base);
        REQUIRE(st.rounds_left() == 7);
        auto lr = st.round_lr(Fe{}, Fe{});
        REQUIRE(lr.first.infinity && !lr.second.infinity);
        REQUIRE(std::equal(lr.second.x.begin(), lr.second.x.end(), &g[8 * (n / 2)]) && std::equal(lr.second.y.begin(), lr.second.y.end(), &g[8 * (n / 2) + 4]));
        bool order = false;
        try { st.round_lr(Fe{}, Fe{}); } catch (const Error& e) { order = e.code == KH_E_INVALID; }
        REQUIRE(order);                                                            // fold must follow L/R
        uint64_t ch = 0x9e3779b97f4a7c15ull;
        auto uu = st.round_fold(ch, ch ^ 0x55);
        REQUIRE(uu.first != uu.second);
        while (st.rounds_left()) { st.round_lr(Fe{}, Fe{}); ch = ch * 6364136223846793005ull + 1; st.round_fold(ch, ~ch); }
        auto fin = st.finish();
        REQUIRE(fin.a0 == fone && fin.b0 == fone && !fin.sg.infinity);
    }
    std::printf("MIRROR_OK\n");
    return 0;
}
