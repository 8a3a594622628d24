// host_sponge.cpp -- the Fiat-Shamir sponges of the Kimchi prover / verifier, host side.
//
// The transcript is the one strictly sequential piece of ProverProof::create (kimchi/src/prover.rs:277-1263): every
// challenge depends on the commitments before it.  It stays on the host next to the caller, exactly as in the reference;
// it is in the library so that a device-resident prover loop (proof_systems_amd/prover.py, or the Rust shim) gets its
// challenges without a Python / FFI big-integer detour in the middle of the timed region.  Restates
//   ArithmeticSponge        poseidon/src/poseidon.rs:60-175, permutation.rs:30-116 (PlonkSpongeConstantsKimchi:
//                           width 3, rate 2, 55 full rounds, x^7, no initial round-key addition, constants.rs:29-41)
//   DefaultFqSponge         poseidon/src/sponge.rs:228-412 (absorb_g / absorb_fq / absorb_fr / challenge / digest)
//   DefaultFrSponge         poseidon/src/sponge.rs:240-282, kimchi/src/plonk_sponge.rs:36-57
// Field elements cross the C ABI as 4 x u64 Montgomery limbs like everywhere else; 128-bit challenges as 2 x u64.
// Checked against the oracle's sponge (itself pinned on poseidon/tests/test_vectors/kimchi.json and on the reference's
// opening-proof bytes) in tests/test_sponge.py -- runs without a GPU.
#include <stdint.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/kimchi_hip.h"
#include "host_ec.hpp"

namespace kh { void set_error(const char* fmt, ...); }

namespace {
#include "poseidon_params.inc"

struct Arith {                       // ArithmeticSponge over field `fid`
    int fid;
    khost::fe s[3];
    bool squeezed = false;           // SpongeState::Squeezed(n) / Absorbed(n)
    int n = 0;
    explicit Arith(int f) : fid(f) { memset(s, 0, sizeof(s)); }
    void permute() {
        khost::Fld F(fid);
        const khost::fe (*mds)[3] = fid == 0 ? POSEIDON_MDS_FP : POSEIDON_MDS_FQ;
        const khost::fe (*rc)[3] = fid == 0 ? POSEIDON_RC_FP : POSEIDON_RC_FQ;
        for (int r = 0; r < 55; r++) {
            khost::fe t[3];
            for (int i = 0; i < 3; i++) {                       // x^7
                const khost::fe x2 = F.sqr(s[i]), x4 = F.sqr(x2);
                t[i] = F.mul(F.mul(x4, x2), s[i]);
            }
            for (int i = 0; i < 3; i++) {
                khost::fe acc = F.mul(mds[i][0], t[0]);
                acc = F.add(acc, F.mul(mds[i][1], t[1]));
                acc = F.add(acc, F.mul(mds[i][2], t[2]));
                s[i] = F.add(acc, rc[r][i]);
            }
        }
    }
    void absorb(const khost::fe& x) {
        khost::Fld F(fid);
        if (!squeezed) {
            if (n == 2) { permute(); s[0] = F.add(s[0], x); n = 1; }
            else { s[n] = F.add(s[n], x); n++; }
        } else {
            s[0] = F.add(s[0], x);
            squeezed = false; n = 1;
        }
    }
    khost::fe squeeze() {
        if (squeezed && n < 2) { n++; return s[n - 1]; }
        permute();
        squeezed = true; n = 1;
        return s[0];
    }
};
}  // namespace

struct kh_sponge {
    int kind;                        // 0: Fq-sponge of `curve`, 1: Fr-sponge of `curve`
    int curve;
    Arith sp;
    std::vector<uint64_t> last_squeezed;
    kh_sponge(int k, int c) : kind(k), curve(c), sp(k == 0 ? khost::base_field_id(c) : khost::scalar_field_id(c)) {}
};

extern "C" {

int kh_sponge_new(int kind, int curve, kh_sponge_t** out) {
    if (!out || (kind != KH_SPONGE_FQ && kind != KH_SPONGE_FR) || (curve != KH_CURVE_VESTA && curve != KH_CURVE_PALLAS)) {
        kh::set_error("kh_sponge_new: bad argument"); return KH_E_INVALID;
    }
    *out = new (std::nothrow) kh_sponge(kind, curve);
    if (!*out) { kh::set_error("out of memory"); return KH_E_NOMEM; }
    return KH_OK;
}
int kh_sponge_clone(const kh_sponge_t* s, kh_sponge_t** out) {
    if (!s || !out) { kh::set_error("kh_sponge_clone: null argument"); return KH_E_INVALID; }
    *out = new (std::nothrow) kh_sponge(*s);
    if (!*out) { kh::set_error("out of memory"); return KH_E_NOMEM; }
    return KH_OK;
}
void kh_sponge_free(kh_sponge_t* s) { delete s; }

// FqSponge::absorb_g: (x, y) of every point, (0, 0) for the point at infinity
int kh_sponge_absorb_g(kh_sponge_t* s, const uint64_t* xy, const uint8_t* inf, size_t n) {
    if (!s || s->kind != KH_SPONGE_FQ || (!xy && n)) { kh::set_error("kh_sponge_absorb_g: needs an Fq-sponge and points"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    for (size_t i = 0; i < n; i++) {
        khost::fe x, y;
        if (inf && inf[i]) { memset(&x, 0, 32); memset(&y, 0, 32); }
        else { memcpy(&x, xy + 8 * i, 32); memcpy(&y, xy + 8 * i + 4, 32); }
        s->sp.absorb(x); s->sp.absorb(y);
    }
    return KH_OK;
}
// elements of the sponge's OWN field (FqSponge::absorb_fq; FrSponge::absorb / absorb_multiple)
int kh_sponge_absorb(kh_sponge_t* s, const uint64_t* x, size_t n) {
    if (!s || (!x && n)) { kh::set_error("kh_sponge_absorb: null argument"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    for (size_t i = 0; i < n; i++) { khost::fe v; memcpy(&v, x + 4 * i, 32); s->sp.absorb(v); }
    return KH_OK;
}
// FqSponge::absorb_fr: SCALAR-field elements into the base-field sponge (sponge.rs:337-366): as one base-field element when
// the scalar modulus is the smaller one (Vesta), else as (high 254 bits, low bit) (Pallas)
int kh_sponge_absorb_fr(kh_sponge_t* s, const uint64_t* x, size_t n) {
    if (!s || s->kind != KH_SPONGE_FQ || (!x && n)) { kh::set_error("kh_sponge_absorb_fr: needs an Fq-sponge"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    khost::Fld SF(khost::scalar_field_id(s->curve)), BF(khost::base_field_id(s->curve));
    const bool scalar_smaller = !khost::geq(SF.f.p, BF.f.p);
    for (size_t i = 0; i < n; i++) {
        khost::fe v; memcpy(&v, x + 4 * i, 32);
        khost::fe c = SF.from_mont(v);                        // canonical integer
        if (scalar_smaller) s->sp.absorb(BF.to_mont(c));
        else {
            khost::fe low = {{c.l[0] & 1, 0, 0, 0}}, high;
            for (int k = 0; k < 4; k++) high.l[k] = (c.l[k] >> 1) | (k < 3 ? c.l[k + 1] << 63 : 0);
            s->sp.absorb(BF.to_mont(high)); s->sp.absorb(BF.to_mont(low));
        }
    }
    return KH_OK;
}
static void squeeze_limbs(kh_sponge* s, size_t k, uint64_t* out) {
    while (s->last_squeezed.size() < k) {
        khost::Fld F(s->sp.fid);
        const khost::fe x = F.from_mont(s->sp.squeeze());
        s->last_squeezed.push_back(x.l[0]); s->last_squeezed.push_back(x.l[1]);      // HIGH_ENTROPY_LIMBS = 2
    }
    for (size_t i = 0; i < k; i++) out[i] = s->last_squeezed[i];
    s->last_squeezed.erase(s->last_squeezed.begin(), s->last_squeezed.begin() + k);
}
// the 128-bit challenge of FqSponge::challenge / FrSponge::challenge (CHALLENGE_LENGTH_IN_LIMBS = 2), raw limbs
int kh_sponge_challenge(kh_sponge_t* s, uint64_t chal[2]) {
    if (!s || !chal) { kh::set_error("kh_sponge_challenge: null argument"); return KH_E_INVALID; }
    squeeze_limbs(s, 2, chal);
    return KH_OK;
}
// the same challenge as an element of the curve's SCALAR field (beta, gamma: used as they are, prover.rs:630-633)
int kh_sponge_challenge_field(kh_sponge_t* s, uint64_t out[4]) {
    if (!s || !out) { kh::set_error("kh_sponge_challenge_field: null argument"); return KH_E_INVALID; }
    uint64_t c[2]; squeeze_limbs(s, 2, c);
    khost::Fld SF(khost::scalar_field_id(s->curve));
    const khost::fe v = {{c[0], c[1], 0, 0}};
    const khost::fe m = SF.to_mont(v);
    memcpy(out, &m, 32);
    return KH_OK;
}
// FqSponge::challenge_fq / digest_fq; FrSponge::digest: one element of the sponge's own field
int kh_sponge_squeeze_field(kh_sponge_t* s, uint64_t out[4]) {
    if (!s || !out) { kh::set_error("kh_sponge_squeeze_field: null argument"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    const khost::fe x = s->sp.squeeze();
    memcpy(out, &x, 32);
    return KH_OK;
}
// FqSponge::digest: the squeezed base-field element as a SCALAR-field element, zero when it does not fit (sponge.rs:368-377)
int kh_sponge_digest(kh_sponge_t* s, uint64_t out[4]) {
    if (!s || !out) { kh::set_error("kh_sponge_digest: null argument"); return KH_E_INVALID; }
    s->last_squeezed.clear();
    if (s->kind == KH_SPONGE_FR) { const khost::fe x = s->sp.squeeze(); memcpy(out, &x, 32); return KH_OK; }
    khost::Fld BF(khost::base_field_id(s->curve)), SF(khost::scalar_field_id(s->curve));
    const khost::fe c = BF.from_mont(s->sp.squeeze());
    khost::fe r; memset(&r, 0, 32);
    if (!khost::geq(c, SF.f.p)) r = SF.to_mont(c);
    memcpy(out, &r, 32);
    return KH_OK;
}

}  // extern "C"
