// kimchi_hip.hpp -- C++ host mirror of the reference's commitment / evaluation-domain
// interface for the hot path, layered on the C ABI (kimchi_hip.h).  The reference is Rust;
// no Rust toolchain exists in the build image, so the host side above the C ABI is C++ with
// the reference's names, argument meaning and error behaviour:
//
//   kimchi_hip::PolyComm                 <- poly_commitment::commitment::PolyComm<G>      (commitment.rs:44-50)
//   kimchi_hip::BlindedCommitment        <- BlindedCommitment<G>                          (commitment.rs:109-116)
//   kimchi_hip::SRS                      <- trait SRS<G> + ipa::SRS<G>                    (lib.rs:61-241, ipa.rs:592-801)
//   kimchi_hip::Radix2EvaluationDomain   <- ark_poly::Radix2EvaluationDomain as used by   kimchi/src/circuits/domains.rs:40-69
//   kimchi_hip::Evaluations::interpolate <- ark_poly Evaluations::interpolate             (kimchi/src/prover.rs:289,377,...)
//   kimchi_hip::DensePolynomial::evaluate_over_domain <- evaluate_over_domain_by_ref      (circuits/constraints.rs:490-495)
//
// Errors: the reference panics (unwrap) on MSM failures and returns
// Err(CommitmentError::BlindersDontMatch) from mask_custom; here every failure throws
// kimchi_hip::Error carrying the C status; BlindersDontMatch is Error with code KH_E_BLINDERS.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "kimchi_hip.h"

namespace kimchi_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) { if (rc != KH_OK) throw Error(rc, kh_last_error()); }

using Fe = std::array<uint64_t, 4>;          // Montgomery limbs: ark-ff's in-memory Fp / Fq
struct Affine {                              // ark_ec short_weierstrass::Affine{x, y, infinity}
    Fe x{}, y{};
    bool infinity = true;
};

struct PolyComm { std::vector<Affine> chunks; };                    // commitment.rs:44-50
struct ScalarPolyComm { std::vector<Fe> chunks; };                  // PolyComm<G::ScalarField> (blinders)
struct BlindedCommitment { PolyComm commitment; ScalarPolyComm blinders; };

enum class Curve : int { Vesta = KH_CURVE_VESTA, Pallas = KH_CURVE_PALLAS };
enum class Field : int { Fp = KH_FIELD_FP, Fq = KH_FIELD_FQ };
inline Field scalar_field(Curve c) { return c == Curve::Vesta ? Field::Fp : Field::Fq; }

struct Radix2EvaluationDomain {              // Radix2EvaluationDomain::new(size): size must be a power of two
    Field field; unsigned log_size_of_group; uint64_t size; Fe group_gen;
    static Radix2EvaluationDomain create(Field f, uint64_t size) {
        if (size == 0 || (size & (size - 1))) throw Error(KH_E_INVALID, "domain size must be a power of two");
        unsigned k = 0; while ((uint64_t(1) << k) < size) k++;
        Radix2EvaluationDomain d{f, k, size, {}};
        check(kh_domain_generator(int(f), k, d.group_gen.data()));
        return d;
    }
};

struct DensePolynomial {                     // coefficients, low degree first
    std::vector<Fe> coeffs;
    // evaluate_over_domain_by_ref(domain): zero-extend to domain.size and forward-NTT (constraints.rs:490-495)
    std::vector<Fe> evaluate_over_domain(const Radix2EvaluationDomain& d) const {
        unsigned k = 0; while ((uint64_t(1) << k) < coeffs.size()) k++;
        if (k > d.log_size_of_group) throw Error(KH_E_INVALID, "polynomial longer than the domain");
        std::vector<Fe> in(size_t(1) << k, Fe{});
        std::copy(coeffs.begin(), coeffs.end(), in.begin());
        std::vector<Fe> out(d.size);
        check(kh_lde(int(d.field), in[0].data(), k, d.log_size_of_group - k, out[0].data(), 1));
        return out;
    }
};

struct Evaluations {                         // natural-order evaluations over `domain`
    std::vector<Fe> evals; Radix2EvaluationDomain domain;
    DensePolynomial interpolate() const {    // Evaluations::interpolate: iNTT including 1/N
        DensePolynomial p{evals};
        p.coeffs.resize(domain.size, Fe{});
        check(kh_ntt(int(domain.field), p.coeffs[0].data(), domain.log_size_of_group, 1, 1));
        return p;
    }
};

class SRS {                                  // trait SRS<G> (lib.rs:61-241) backed by device-resident bases
    kh_srs_t* h_ = nullptr; Curve curve_; size_t n_ = 0;
    static std::vector<uint64_t> pack(const std::vector<Fe>& v) { std::vector<uint64_t> o(4 * v.size()); for (size_t i = 0; i < v.size(); i++) std::copy(v[i].begin(), v[i].end(), o.begin() + 4 * i); return o; }
    static PolyComm unpack(const std::vector<uint64_t>& xy, const std::vector<uint8_t>& inf, size_t cnt) {
        PolyComm pc; pc.chunks.resize(cnt);
        for (size_t j = 0; j < cnt; j++) { pc.chunks[j].infinity = inf[j] != 0; if (!inf[j]) { std::copy(&xy[8 * j], &xy[8 * j + 4], pc.chunks[j].x.begin()); std::copy(&xy[8 * j + 4], &xy[8 * j + 8], pc.chunks[j].y.begin()); } }
        return pc;
    }
  public:
    SRS(Curve c, const std::vector<std::array<uint64_t, 8>>& g) : curve_(c), n_(g.size()) { check(kh_srs_create(int(c), g[0].data(), g.size(), &h_)); }
    // SRS::create(depth) (ipa.rs:751-778)
    static SRS create(Curve c, size_t depth) {
        std::vector<std::array<uint64_t, 8>> g(depth);
        check(kh_srs_generate(int(c), 0, depth, g[0].data(), 0));
        return SRS(c, g);
    }
    SRS(const SRS&) = delete; SRS& operator=(const SRS&) = delete;
    SRS(SRS&& o) noexcept : h_(o.h_), curve_(o.curve_), n_(o.n_) { o.h_ = nullptr; }
    ~SRS() { kh_srs_free(h_); }

    size_t max_poly_size() const { return n_; }
    size_t size() const { return n_; }
    Affine blinding_commitment() const { uint64_t h[8]; check(kh_srs_get_blinding_base(h_, h)); Affine a; a.infinity = false; std::copy(h, h + 4, a.x.begin()); std::copy(h + 4, h + 8, a.y.begin()); return a; }

    // registers SRS::get_lagrange_basis(domain) chunk-wise (ipa.rs:780-801); computed by ipa.rs:1065-1172
    void set_lagrange_basis(const Radix2EvaluationDomain& d, unsigned chunk, const std::vector<Affine>& basis) {
        std::vector<uint64_t> xy(8 * basis.size()); std::vector<uint8_t> inf(basis.size());
        for (size_t i = 0; i < basis.size(); i++) { inf[i] = basis[i].infinity; std::copy(basis[i].x.begin(), basis[i].x.end(), &xy[8 * i]); std::copy(basis[i].y.begin(), basis[i].y.end(), &xy[8 * i + 4]); }
        check(kh_srs_set_lagrange(h_, d.log_size_of_group, chunk, xy.data(), inf.data(), basis.size()));
    }

    PolyComm commit_non_hiding(const DensePolynomial& plnm, size_t num_chunks) const {          // ipa.rs:638-683
        size_t cap = std::max<size_t>(std::max<size_t>(num_chunks, (plnm.coeffs.size() + n_ - 1) / n_), 1);
        std::vector<uint64_t> xy(8 * cap); std::vector<uint8_t> inf(cap); size_t cnt = 0;
        auto c = pack(plnm.coeffs);
        check(kh_commit_non_hiding(h_, c.data(), plnm.coeffs.size(), num_chunks, xy.data(), inf.data(), &cnt));
        return unpack(xy, inf, cnt);
    }
    PolyComm commit_evaluations_non_hiding(const Radix2EvaluationDomain& domain, const Evaluations& plnm) const {   // ipa.rs:706-728
        int chunks = kh_srs_lagrange_chunks(h_, domain.log_size_of_group);
        size_t cap = chunks > 0 ? size_t(chunks) : 1;
        std::vector<uint64_t> xy(8 * cap); std::vector<uint8_t> inf(cap); size_t cnt = 0;
        auto e = pack(plnm.evals);
        check(kh_commit_evaluations_non_hiding(h_, domain.log_size_of_group, e.data(), plnm.evals.size(), xy.data(), inf.data(), &cnt));
        return unpack(xy, inf, cnt);
    }
    BlindedCommitment mask_custom(const PolyComm& com, const ScalarPolyComm& blinders) const {   // ipa.rs:605-622
        std::vector<uint64_t> xy(8 * com.chunks.size()), out(8 * com.chunks.size()); std::vector<uint8_t> inf(com.chunks.size()), oinf(com.chunks.size());
        for (size_t j = 0; j < com.chunks.size(); j++) { inf[j] = com.chunks[j].infinity; std::copy(com.chunks[j].x.begin(), com.chunks[j].x.end(), &xy[8 * j]); std::copy(com.chunks[j].y.begin(), com.chunks[j].y.end(), &xy[8 * j + 4]); }
        auto b = pack(blinders.chunks);
        check(kh_mask_custom(h_, xy.data(), inf.data(), com.chunks.size(), b.data(), blinders.chunks.size(), out.data(), oinf.data()));
        return BlindedCommitment{unpack(out, oinf, com.chunks.size()), blinders};
    }
    BlindedCommitment commit_custom(const DensePolynomial& plnm, size_t num_chunks, const ScalarPolyComm& blinders) const {   // ipa.rs:686-693
        return mask_custom(commit_non_hiding(plnm, num_chunks), blinders);
    }
    BlindedCommitment commit_evaluations_custom(const Radix2EvaluationDomain& d, const Evaluations& plnm, const ScalarPolyComm& blinders) const {   // ipa.rs:740-748
        return mask_custom(commit_evaluations_non_hiding(d, plnm), blinders);
    }
    kh_srs_t* raw() const { return h_; }
};

// The folding loop of SRS::open (ipa.rs:929-1018) on device-resident vectors.  The caller owns the sponge and the RNG:
//   OpeningRounds st(srs, p.coeffs, b_init, u_base);
//   while (st.rounds_left()) { auto [l, r] = st.round_lr(rand_l, rand_r); absorb(l, r); auto [u, u_inv] = st.round_fold(squeeze()); }
//   auto fin = st.finish();   // a0, b0, sg
class OpeningRounds {
    kh_ipa_t* st_ = nullptr;
    static Affine point(const uint64_t* xy, uint8_t inf) { Affine a; a.infinity = inf != 0; if (!inf) { std::copy(xy, xy + 4, a.x.begin()); std::copy(xy + 4, xy + 8, a.y.begin()); } return a; }
public:
    struct Final { Fe a0, b0; Affine sg; };
    OpeningRounds(const SRS& srs, const std::vector<Fe>& a, const std::vector<Fe>& b, const Affine& u_base) {
        std::vector<uint64_t> av(4 * a.size()), bv(4 * b.size());
        for (size_t i = 0; i < a.size(); i++) std::copy(a[i].begin(), a[i].end(), &av[4 * i]);
        for (size_t i = 0; i < b.size(); i++) std::copy(b[i].begin(), b[i].end(), &bv[4 * i]);
        uint64_t u[8]; std::copy(u_base.x.begin(), u_base.x.end(), u); std::copy(u_base.y.begin(), u_base.y.end(), u + 4);
        check(kh_ipa_begin(srs.raw(), av.data(), a.size(), bv.data(), b.size(), u, &st_));
    }
    OpeningRounds(const OpeningRounds&) = delete; OpeningRounds& operator=(const OpeningRounds&) = delete;
    ~OpeningRounds() { kh_ipa_free(st_); }
    int rounds_left() const { return kh_ipa_rounds_left(st_); }
    std::pair<Affine, Affine> round_lr(const Fe& rand_l, const Fe& rand_r) {                      // ipa.rs:940-961
        uint64_t xy[16]; uint8_t inf[2];
        check(kh_ipa_round_lr(st_, rand_l.data(), rand_r.data(), xy, inf));
        return {point(xy, inf[0]), point(xy + 8, inf[1])};
    }
    // u_pre: the 128-bit prechallenge (low limb first); returns (u, u^-1) = (u_pre.to_field(endo_r), its inverse), ipa.rs:972-978
    std::pair<Fe, Fe> round_fold(uint64_t u_pre_lo, uint64_t u_pre_hi) {
        const uint64_t c[2] = {u_pre_lo, u_pre_hi}; Fe u, ui;
        check(kh_ipa_round_fold(st_, c, u.data(), ui.data()));
        return {u, ui};
    }
    Final finish() {                                                                                 // ipa.rs:1009-1018
        Final f; uint64_t sg[8]; uint8_t inf = 0;
        check(kh_ipa_finish(st_, f.a0.data(), f.b0.data(), sg, &inf));
        f.sg = point(sg, inf);
        return f;
    }
};

}  // namespace kimchi_hip
