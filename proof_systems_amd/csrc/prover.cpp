// prover.cpp -- ProverProof::create (kimchi/src/prover.rs:187-1515) as a native host loop over this library's own C ABI.
//
// The same protocol proof_systems_amd/prover.py runs from Python, for everything create_recursive takes (previous challenges, lookups, runtime tables), written against the public
// entry points only (kh_ntt_dev, kh_gate_evaluations_dev, kh_msm_submit, kh_ipa_open, kh_sponge_*, ...): a Rust or C caller gets a whole proof
// with one call, no interpreter in the measured latency, and several prover threads do not share a GIL.  Every device step is the entry
// point the Python prover calls at the same place, with the same arguments, so the two give the same proof for the same randomness
// (tests/test_gpu_native_prover.py compares them field element for field element; the Python prover is pinned on the reference's whole-proof
// vector through the oracle prover).  Host arithmetic: khost::Fld on Montgomery limbs (the wire form).
//
//   round 1  prover.rs:254-327   zero-knowledge rows, public polynomial, 15 witness commitments (one batched MSM over the Lagrange basis)
//   round 2  prover.rs:676-707   beta, gamma; permutation aggregation z (permutation.rs:510-568), commitment
//   round 3  prover.rs:709-985   alpha; generic + permutation + gate constraints on d8, division by Z_H, boundary quotients, t commitment
//   round 4  prover.rs:987-1263  zeta; chunked evaluations, ft (Maller), Fr-sponge -> v, u
//   round 5  prover.rs:1265-1500 SRS::open over (public, ft, z, selectors, w, coefficients, sigma, optional selectors)
#include <stdint.h>
#include <string.h>
#include <sys/random.h>
#include <chrono>
#include <new>
#include <vector>

#include "../../include/kimchi_hip.h"
#include "host_ec.hpp"

namespace kh { void set_error(const char* fmt, ...); }

namespace {
using khost::fe;
constexpr size_t COLUMNS = 15, PERMUTS = 7, SEL0 = COLUMNS + 2 + PERMUTS, OPT0 = SEL0 + 5;
constexpr int ALPHA_PERM0 = 21;                      // the gates take the first 21 powers of alpha (linearization.rs:56-58), the permutation the next 3
const char* const LIB_GATES[5] = {"Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar"};

struct Dev {                                         // a device allocation that lives as long as the proof is being made
    uint64_t* p = nullptr;
    Dev() = default;
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    ~Dev() { if (p) (void)kh_dev_free(p); }
    int alloc(size_t elems) { return kh_dev_alloc((void**)&p, elems * 32); }
    uint64_t* at(size_t elem) const { return p + 4 * elem; }
};
struct SpongeH {
    kh_sponge_t* s = nullptr;
    ~SpongeH() { if (s) kh_sponge_free(s); }
};
fe load(const uint64_t* l) { fe r; memcpy(&r, l, 32); return r; }
fe fpow(const khost::Fld& F, fe base, uint64_t e) {
    fe acc = F.f.one;
    while (e) { if (e & 1) acc = F.mul(acc, base); base = F.sqr(base); e >>= 1; }
    return acc;
}
// sum_k c[k] x^k over `cnt` consecutive elements (ProofEvaluations::combine: the chunks of one evaluation)
fe horner(const khost::Fld& F, const fe* c, size_t cnt, const fe& x) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t k = cnt; k-- > 0;) acc = F.add(F.mul(acc, x), c[k]);
    return acc;
}
int os_random(int fid, size_t k, fe* out) {          // uniform elements of the field, used as Montgomery limbs
    const fe& p = khost::field(fid).p;
    for (size_t i = 0; i < k;) {
        fe buf[8];
        if (getrandom(buf, sizeof(buf), 0) != (ssize_t)sizeof(buf)) { kh::set_error("getrandom failed"); return KH_E_DEVICE; }
        for (int j = 0; j < 8 && i < k; j++) {
            buf[j].l[3] &= 0x7fffffffffffffffULL;
            if (!khost::geq(buf[j], p)) out[i++] = buf[j];
        }
    }
    return KH_OK;
}

// ---- the lookup argument (kimchi/src/circuits/lookup/): protocol data and the expressions of its constraints as token programs ----
// LookupPattern::lookups (lookups.rs:417-487): per pattern the joint lookups of a row -- table id (a constant, or a witness column) and the
// witness columns of the entry.  Pattern ids: 0 Xor, 1 Lookup, 2 RangeCheck, 3 ForeignFieldMul (the reference's order).
struct JointLookup { int tid_is_column, tid, ncell, cells[3]; };
struct Pattern { int n; JointLookup l[4]; };
const Pattern PATTERNS[4] = {
    {4, {{0, 0, 3, {3, 7, 11}}, {0, 0, 3, {4, 8, 12}}, {0, 0, 3, {5, 9, 13}}, {0, 0, 3, {6, 10, 14}}}},
    {3, {{1, 0, 2, {1, 2, 0}}, {1, 0, 2, {3, 4, 0}}, {1, 0, 2, {5, 6, 0}}, {0, 0, 0, {0, 0, 0}}}},
    {4, {{0, 1, 1, {3, 0, 0}}, {0, 1, 1, {4, 0, 0}}, {0, 1, 1, {5, 0, 0}}, {0, 1, 1, {6, 0, 0}}}},
    {4, {{0, 1, 1, {7, 0, 0}}, {0, 1, 1, {8, 0, 0}}, {0, 1, 1, {9, 0, 0}}, {0, 1, 1, {10, 0, 0}}}},
};
// a postfix token program under construction (KH_TOK_*), constants interned by value
struct Prog {
    std::vector<uint32_t> t;
    std::vector<fe> consts;
    void push(uint32_t op, uint32_t a) { t.push_back(op); t.push_back(a); }
    void C(const fe& v) {
        for (size_t i = 0; i < consts.size(); i++) if (khost::eq(consts[i], v)) { push(KH_TOK_CONST, (uint32_t)i); return; }
        consts.push_back(v); push(KH_TOK_CONST, (uint32_t)(consts.size() - 1));
    }
    void cell(uint32_t col, int next = 0) { push(KH_TOK_CELL, 2 * col + (next ? 1u : 0u)); }
    void add() { push(KH_TOK_ADD, 0); }
    void sub() { push(KH_TOK_SUB, 0); }
    void mul() { push(KH_TOK_MUL, 0); }
    int run(int fid, const std::vector<const uint64_t*>& cols, const std::vector<size_t>& lens, size_t rows, unsigned stride, unsigned next_shift, int accumulate, uint64_t* out) const {
        return kh_expr_evaluations_dev(fid, t.data(), t.size() / 2, cols.data(), lens.data(), cols.size(), (const uint64_t*)consts.data(), consts.size(), rows, stride,
                                       next_shift, accumulate, out);
    }
};
struct LookupChallenges { fe jc, tic, beta, gamma, gb1; fe prefactor[5]; };   // prefactor[k] = (gamma + dummy)^k (1 + beta)^max_per_row, dummy = 0
// combine_table_entry (tables/mod.rs:147-162) of one joint lookup: Horner in the joint combiner from the last cell + table_id_combiner * id
void emit_joint(Prog& p, const khost::Fld& F, const JointLookup& L, const LookupChallenges& ch) {
    p.cell((uint32_t)L.cells[L.ncell - 1]);
    for (int i = L.ncell - 2; i >= 0; i--) { p.C(ch.jc); p.mul(); p.cell((uint32_t)L.cells[i]); p.add(); }
    if (L.tid_is_column) { p.cell((uint32_t)L.tid); p.C(ch.tic); p.mul(); p.add(); }
    else if (L.tid) { fe id = {{(uint64_t)L.tid, 0, 0, 0}}; p.C(F.mul(ch.tic, F.to_mont(id))); p.add(); }
}
// (1 + beta)^max_per_row (gamma + dummy)^padding prod (gamma + joint value)   (constraints.rs:497-523)
void emit_fterm(Prog& p, const khost::Fld& F, const Pattern* pat, size_t mpr, const LookupChallenges& ch) {
    const int n = pat ? pat->n : 0;
    p.C(ch.prefactor[mpr - (size_t)n]);
    for (int i = 0; i < n; i++) { p.C(ch.gamma); emit_joint(p, F, pat->l[i], ch); p.add(); p.mul(); }
}
// numerator of an aggregation row: f_chunk * t_chunk, with the pattern selectors at columns sel0.., the combined table at column `table`
void emit_numerator(Prog& p, const khost::Fld& F, const std::vector<int>& pats, size_t mpr, const LookupChallenges& ch, uint32_t sel0, uint32_t table) {
    p.C(F.f.one);
    for (size_t k = 0; k < pats.size(); k++) { p.cell(sel0 + (uint32_t)k); if (k) p.add(); }
    p.sub();                                                            // 1 - sum of the selectors: a row without lookups
    emit_fterm(p, F, nullptr, mpr, ch); p.mul();
    for (size_t k = 0; k < pats.size(); k++) { p.cell(sel0 + (uint32_t)k); emit_fterm(p, F, &PATTERNS[pats[k]], mpr, ch); p.mul(); p.add(); }
    p.C(ch.gb1); p.cell(table); p.add(); p.C(ch.beta); p.cell(table, 1); p.mul(); p.add();      // t_chunk = gamma (1 + beta) + t + beta t'
    p.mul();
}
// denominator: prod_i (gamma (1 + beta) + s_i + beta s_i') with the roles of s_i, s_i' swapped for odd i (the snake)
void emit_denominator(Prog& p, size_t mpr, const LookupChallenges& ch, uint32_t sorted0) {
    for (size_t i = 0; i <= mpr; i++) {
        const int odd = (int)(i & 1);
        p.C(ch.gb1); p.cell(sorted0 + (uint32_t)i, odd); p.add(); p.C(ch.beta); p.cell(sorted0 + (uint32_t)i, !odd); p.mul(); p.add();
        if (i) p.mul();
    }
}
}  // namespace

struct kh_lookup_index {
    std::vector<int> pats;                           // pattern ids present, in the reference's order
    std::vector<const uint64_t*> sel1, selc, sel8, tcols;
    const uint64_t* tids = nullptr;
    const uint64_t* atoms8[3] = {nullptr, nullptr, nullptr};
    size_t mpr = 0, mjs = 0;
    // runtime tables (lookup/runtime_tables.rs): the rows of the combined table whose second column arrives with each proof
    const uint64_t *rtsel1 = nullptr, *rtselc = nullptr, *rtsel8 = nullptr;
    size_t rt_offset = 0, rt_len = 0;
};

struct kh_prover_index {
    kh_srs_t* srs = nullptr;
    int curve = 0, fid = 0;
    unsigned logn = 0, live = 0;
    size_t n = 0, size = 0, nch = 1, zk = 3, pub = 0, ncol = 0;
    const uint64_t *d1 = nullptr, *dc = nullptr, *d8 = nullptr;
    std::vector<int> optional;                       // kh gate ids of the optional selector columns
    int lib_gate[5] = {0, 0, 0, 0, 0}, gid_generic = -1, gid_perm = -1;
    fe shifts[7], digest, omega, endo;
    uint64_t* zero_poly = nullptr;                   // n zeros on the device: the public polynomial of a circuit without public inputs
    std::vector<uint64_t> zsel_xy; std::vector<uint8_t> zsel_inf;   // commitment to the zero polynomial masked with 1 (= h per chunk)
    kh_lookup_index* lk = nullptr;                   // kh_prover_index_attach_lookup
    const uint64_t* col1(size_t k) const { return d1 + 4 * k * n; }
    const uint64_t* colc(size_t k) const { return dc + 4 * k * n; }
    const uint64_t* col8(size_t k) const { return d8 + 4 * k * 8 * n; }
};

struct kh_proof {
    struct Sec { std::vector<uint64_t> limbs; std::vector<uint8_t> flags; size_t count = 0; bool points = false; };
    Sec sec[15];
    double phase[6] = {0, 0, 0, 0, 0, 0};
    void set_points(int s, const uint64_t* xy, const uint8_t* inf, size_t cnt) {
        sec[s].limbs.assign(xy, xy + 8 * cnt); sec[s].flags.assign(inf, inf + cnt); sec[s].count = cnt; sec[s].points = true;
    }
    void set_elems(int s, const fe* v, size_t cnt) {
        sec[s].limbs.resize(4 * cnt); if (cnt) memcpy(sec[s].limbs.data(), v, 32 * cnt); sec[s].flags.clear(); sec[s].count = cnt; sec[s].points = false;
    }
};

extern "C" {

#define KP(expr)                                   \
    do {                                           \
        int rc_ = (expr);                          \
        if (rc_ != KH_OK) { (void)kh_sync(); return rc_; }   \
    } while (0)
#define KP_REQUIRE(cond, ...)                                        \
    do {                                                             \
        if (!(cond)) { kh::set_error(__VA_ARGS__); (void)kh_sync(); return KH_E_INVALID; } \
    } while (0)

int kh_prover_index_new(kh_srs_t* srs, unsigned log2_n, unsigned zk_rows, unsigned public_inputs, const uint64_t* d1_dev, const uint64_t* dc_dev,
                        const uint64_t* d8_dev, const int* optional_gates, size_t n_optional, unsigned live_mask, const uint64_t* shifts,
                        const uint64_t digest[4], kh_prover_index_t** out) {
    if (!srs || !d1_dev || !dc_dev || !d8_dev || !shifts || !digest || !out || (n_optional && !optional_gates) || log2_n > 26) {
        kh::set_error("kh_prover_index_new: bad argument"); return KH_E_INVALID;
    }
    kh_prover_index* ix = new (std::nothrow) kh_prover_index();
    if (!ix) { kh::set_error("out of memory"); return KH_E_NOMEM; }
    ix->srs = srs; ix->curve = kh_srs_curve(srs); ix->fid = ix->curve == KH_CURVE_VESTA ? KH_FIELD_FP : KH_FIELD_FQ;
    ix->logn = log2_n; ix->n = (size_t)1 << log2_n; ix->size = kh_srs_size(srs);
    ix->nch = ix->n < ix->size ? 1 : ix->n / ix->size;
    ix->zk = zk_rows; ix->pub = public_inputs; ix->live = live_mask;
    ix->d1 = d1_dev; ix->dc = dc_dev; ix->d8 = d8_dev;
    ix->optional.assign(optional_gates, optional_gates + n_optional);
    ix->ncol = OPT0 + n_optional;
    int rc = KH_OK;
    auto fail = [&](int code) { kh_prover_index_free(ix); return code; };
    if (zk_rows <= (2 * (PERMUTS + 1) * ix->nch - 2) / PERMUTS || zk_rows >= ix->n) { kh::set_error("NotZeroKnowledge: zk_rows %u for %zu chunks", zk_rows, ix->nch); return fail(KH_E_INVALID); }
    if (kh_srs_lagrange_chunks(srs, log2_n) == 0) { kh::set_error("the Lagrange basis of 2^%u is not registered on this SRS (kh_srs_compute_lagrange)", log2_n); return fail(KH_E_NOTFOUND); }
    const int ngates = kh_gate_count();
    for (int g = 0; g < ngates; g++) {
        const char* nm = kh_gate_name(g);
        for (int k = 0; k < 5; k++) if (!strcmp(nm, LIB_GATES[k])) ix->lib_gate[k] = g;
        if (!strcmp(nm, "Generic")) ix->gid_generic = g;
        if (!strcmp(nm, "Permutation")) ix->gid_perm = g;
    }
    for (int g : ix->optional) if (g < 0 || g >= ngates) { kh::set_error("unknown optional gate id %d", g); return fail(KH_E_INVALID); }
    for (int i = 0; i < 7; i++) ix->shifts[i] = load(shifts + 4 * i);
    ix->digest = load(digest);
    uint64_t w[4], eq[4], er[4];
    if ((rc = kh_domain_generator(ix->fid, log2_n, w))) return fail(rc);
    ix->omega = load(w);
    if ((rc = kh_endos(1 - ix->curve, eq, er))) return fail(rc);          // VerifierIndex::endo = endos::<OtherCurve>().0: an element of this scalar field
    ix->endo = load(eq);
    if ((rc = kh_dev_alloc((void**)&ix->zero_poly, ix->n * 32))) return fail(rc);
    if ((rc = kh_dev_memset_zero(ix->zero_poly, ix->n * 32))) return fail(rc);
    // the commitment to the zero public polynomial: infinity per chunk (commit_non_hiding), masked with blinders 1 (prover.rs:296-309)
    const khost::Fld F(ix->fid);
    std::vector<uint64_t> zxy(8 * ix->nch, 0), ones(4 * ix->nch); std::vector<uint8_t> zinf(ix->nch, 1);
    for (size_t c = 0; c < ix->nch; c++) memcpy(&ones[4 * c], &F.f.one, 32);
    ix->zsel_xy.resize(8 * ix->nch); ix->zsel_inf.resize(ix->nch);
    if ((rc = kh_mask_custom(srs, zxy.data(), zinf.data(), ix->nch, ones.data(), ix->nch, ix->zsel_xy.data(), ix->zsel_inf.data()))) return fail(rc);
    if ((rc = kh_sync())) return fail(rc);
    *out = ix;
    return KH_OK;
}
void kh_prover_index_free(kh_prover_index_t* ix) {
    if (!ix) return;
    if (ix->zero_poly) (void)kh_dev_free(ix->zero_poly);
    delete ix->lk;
    delete ix;
}
int kh_prover_index_attach_lookup(kh_prover_index_t* ix, const int* patterns, size_t n_patterns, const uint64_t* const* selectors_d1,
                                  const uint64_t* const* selectors_c, const uint64_t* const* selectors_d8, const uint64_t* const* table_cols_d1, size_t n_table_cols,
                                  const uint64_t* table_ids_d1, const uint64_t* const* atoms_d8) {
    if (!ix || !patterns || !n_patterns || n_patterns > 4 || !selectors_d1 || !selectors_c || !selectors_d8 || !table_cols_d1 || !n_table_cols || !atoms_d8) {
        kh::set_error("kh_prover_index_attach_lookup: bad argument"); return KH_E_INVALID;
    }
    kh_lookup_index* lk = new (std::nothrow) kh_lookup_index();
    if (!lk) { kh::set_error("out of memory"); return KH_E_NOMEM; }
    for (size_t k = 0; k < n_patterns; k++) {
        if (patterns[k] < 0 || patterns[k] > 3 || (k && patterns[k] <= patterns[k - 1]) || !selectors_d1[k] || !selectors_c[k] || !selectors_d8[k]) {
            kh::set_error("lookup patterns must be distinct ids 0..3 in increasing order, with their selector columns"); delete lk; return KH_E_INVALID;
        }
        lk->pats.push_back(patterns[k]);
        lk->sel1.push_back(selectors_d1[k]); lk->selc.push_back(selectors_c[k]); lk->sel8.push_back(selectors_d8[k]);
        const Pattern& P = PATTERNS[patterns[k]];
        if ((size_t)P.n > lk->mpr) lk->mpr = (size_t)P.n;
        for (int i = 0; i < P.n; i++) if ((size_t)P.l[i].ncell > lk->mjs) lk->mjs = (size_t)P.l[i].ncell;
    }
    for (size_t k = 0; k < n_table_cols; k++) { if (!table_cols_d1[k]) { kh::set_error("null table column"); delete lk; return KH_E_INVALID; } lk->tcols.push_back(table_cols_d1[k]); }
    lk->tids = table_ids_d1;
    for (int a = 0; a < 3; a++) { if (!atoms_d8[a]) { kh::set_error("null atom column"); delete lk; return KH_E_INVALID; } lk->atoms8[a] = atoms_d8[a]; }
    delete ix->lk;
    ix->lk = lk;
    return KH_OK;
}
int kh_prover_index_attach_runtime_tables(kh_prover_index_t* ix, const uint64_t* selector_d1, const uint64_t* selector_c, const uint64_t* selector_d8, size_t offset,
                                          size_t length) {
    if (!ix || !ix->lk || !selector_d1 || !selector_c || !selector_d8 || !length || offset + length + ix->zk >= ix->n || ix->lk->tcols.size() < 2) {
        kh::set_error("kh_prover_index_attach_runtime_tables: attach the lookup index first; %zu runtime rows at %zu must fit the table (two columns at least)", length, offset);
        return KH_E_INVALID;
    }
    ix->lk->rtsel1 = selector_d1; ix->lk->rtselc = selector_c; ix->lk->rtsel8 = selector_d8;
    ix->lk->rt_offset = offset; ix->lk->rt_len = length;
    return KH_OK;
}

size_t kh_prove_randomness_count(const kh_prover_index_t* ix, int witness_on_host) {
    if (!ix) return 0;
    size_t logs = 0; while (((size_t)1 << logs) < ix->size) logs++;
    size_t lookups = ix->lk ? (ix->lk->mpr + 1) * (ix->zk + ix->nch) + ix->zk + ix->nch : 0;   // sorted columns: zk rows + blinders; aggregation: zk rows + blinders
    if (ix->lk && ix->lk->rtsel1) lookups += ix->zk + ix->nch;                               // the runtime table column: zk rows + blinders
    return (witness_on_host ? COLUMNS * ix->zk : 0) + COLUMNS * ix->nch + lookups + 2 + ix->nch + 7 * ix->nch + 2 * logs + 2;
}

int kh_prove(kh_prover_index_t* ix, const uint64_t* witness, size_t rows, const uint64_t* witness_dev, const uint64_t* randomness, size_t n_random,
             unsigned flags, kh_proof_t** out) {
    return kh_prove_full(ix, witness, rows, witness_dev, randomness, n_random, flags, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, out);
}
int kh_prove_recursive(kh_prover_index_t* ix, const uint64_t* witness, size_t rows, const uint64_t* witness_dev, const uint64_t* randomness, size_t n_random,
                       unsigned flags, const uint64_t* prev_chals, const unsigned* prev_rounds, const uint64_t* prev_comm_xy, const uint8_t* prev_comm_inf,
                       const size_t* prev_comm_chunks, size_t n_prev, kh_proof_t** out) {
    return kh_prove_full(ix, witness, rows, witness_dev, randomness, n_random, flags, prev_chals, prev_rounds, prev_comm_xy, prev_comm_inf, prev_comm_chunks, n_prev,
                         nullptr, 0, out);
}

int kh_prove_full(kh_prover_index_t* ix, const uint64_t* witness, size_t rows, const uint64_t* witness_dev, const uint64_t* randomness, size_t n_random,
                  unsigned flags, const uint64_t* prev_chals, const unsigned* prev_rounds, const uint64_t* prev_comm_xy, const uint8_t* prev_comm_inf,
                  const size_t* prev_comm_chunks, size_t n_prev, const uint64_t* runtime_values, size_t n_runtime, kh_proof_t** out) {
    if (ix && ((ix->lk && ix->lk->rtsel1) ? (!runtime_values || n_runtime != ix->lk->rt_len) : n_runtime != 0)) {
        kh::set_error("RuntimeTablesInconsistent: the index has %zu runtime table rows, the proof brings %zu", (ix->lk && ix->lk->rtsel1) ? ix->lk->rt_len : (size_t)0, n_runtime);
        return KH_E_INVALID;
    }
    if (!ix || !out || (!witness == !witness_dev)) { kh::set_error("kh_prove: give the witness either on the host or on the device"); return KH_E_INVALID; }
    if (n_prev && (!prev_chals || !prev_rounds || !prev_comm_xy || !prev_comm_inf || !prev_comm_chunks)) { kh::set_error("kh_prove_recursive: null previous-challenge argument"); return KH_E_INVALID; }
    const bool check = flags & KH_PROVE_CHECK, all_gates = flags & KH_PROVE_ALL_GATES;
    static const bool eager_env = getenv("KH_PROVE_EAGER_CHECK") && atoi(getenv("KH_PROVE_EAGER_CHECK")) != 0;
    const bool eager = check && ((flags & KH_PROVE_EAGER_CHECK) || eager_env);       // fail at the phase the reference fails at (a stream stall per check)
    const int fid = ix->fid, curve = ix->curve;
    const unsigned logn = ix->logn;
    const size_t n = ix->n, size = ix->size, nch = ix->nch, zk = ix->zk, nopt = ix->optional.size();
    kh_srs_t* srs = ix->srs;
    const khost::Fld F(fid);
    const fe one = F.f.one, zero = {{0, 0, 0, 0}};
    struct DeviceRestore { int prev; ~DeviceRestore() { if (prev >= 0) (void)kh_set_device(prev); } } device_restore{kh_get_device()};
    KP(kh_set_device(kh_srs_device(srs)));           // the index lives on the SRS's device: this thread works there until the proof is made
    // a context of this thread's own for the proof (own main stream, pipeline slots, lock): provers on several threads do not queue behind each other
    struct PrivateContext {
        bool mine = false;
        ~PrivateContext() { if (mine) (void)kh_private_context_end(); }
    } private_context;
    if (!(flags & KH_PROVE_SHARED_CONTEXT) && !kh_private_context_active()) {
        KP(kh_private_context_begin()); private_context.mine = true;
        static const bool keep_timers = getenv("KH_PROVE_TIMERS") && atoi(getenv("KH_PROVE_TIMERS")) != 0;
        KP(kh_set_phase_timers(keep_timers ? 1 : 0));    // (this context is ours for the call: no per-phase events between the kernels of the proof; a pooled context may come with them on)
    }
    // ---- the randomness of the whole proof, in the reference's draw order
    const size_t need = kh_prove_randomness_count(ix, witness != nullptr);
    std::vector<fe> rnd(need + 64, fe{{0, 0, 0, 0}});   // (slack: a miscounted draw reads zeros, and the count check at the end reports it)
    if (randomness) {
        KP_REQUIRE(n_random == need, "kh_prove: %zu random elements given, %zu drawn (kh_prove_randomness_count)", n_random, need);
        memcpy(rnd.data(), randomness, 32 * need);
    } else KP(os_random(fid, need, rnd.data()));
    size_t rpos = 0;
    auto draw = [&](size_t k) { const fe* p = rnd.data() + rpos; rpos += k; return p; };
    kh_proof* pr = new (std::nothrow) kh_proof();
    if (!pr) { kh::set_error("out of memory"); return KH_E_NOMEM; }
    struct Guard { kh_proof* p; ~Guard() { delete p; } } guard{pr};
    auto t_prev = std::chrono::steady_clock::now();
    int phase_i = 0;
    auto mark = [&]() { auto t = std::chrono::steady_clock::now(); pr->phase[phase_i++] = std::chrono::duration<double>(t - t_prev).count(); t_prev = t; };
    const size_t NB = n, N8 = 8 * n;                 // elements per d1 / d8 column
    // commitments: chunk lists, flat (commitment after commitment)
    auto commit_evals = [&](const uint64_t* ptr, size_t k, std::vector<uint64_t>& xy, std::vector<uint8_t>& inf) -> int {
        // SRS::commit_evaluations_non_hiding of k columns: per chunk of the Lagrange basis one batched MSM over all n evaluations
        xy.assign(8 * k * nch, 0); inf.assign(k * nch, 0);
        std::vector<uint64_t> o(8 * k); std::vector<uint8_t> oi(k);
        for (size_t c = 0; c < nch; c++) {
            int rc = kh_msm_batch_dev(srs, (int)logn, (unsigned)c, 0, ptr, n, k, 1, o.data(), oi.data());
            if (rc) return rc;
            for (size_t i = 0; i < k; i++) { memcpy(&xy[8 * (i * nch + c)], &o[8 * i], 64); inf[i * nch + c] = oi[i]; }
        }
        return KH_OK;
    };
    auto commit_coeffs = [&](const uint64_t* ptr, size_t length, size_t chunks, std::vector<uint64_t>& xy, std::vector<uint8_t>& inf) -> int {
        // SRS::commit_non_hiding (ipa.rs:638-683): chunks of the SRS size, padded with the point at infinity
        size_t cnt = (length + size - 1) / size; if (cnt < chunks) cnt = chunks; if (cnt < 1) cnt = 1;
        xy.assign(8 * cnt, 0); inf.assign(cnt, 1);
        const size_t full = length / size, rem = length - full * size;
        if (full) { int rc = kh_msm_batch_dev(srs, KH_BASIS_G, 0, 0, ptr, size, full, 1, xy.data(), inf.data()); if (rc) return rc; }
        if (rem) { int rc = kh_msm_batch_dev(srs, KH_BASIS_G, 0, 0, ptr + 4 * full * size, rem, 1, 1, &xy[8 * full], &inf[full]); if (rc) return rc; }
        return KH_OK;
    };
    auto mask = [&](const std::vector<uint64_t>& xy, const std::vector<uint8_t>& inf, const fe* blinders, std::vector<uint64_t>& oxy, std::vector<uint8_t>& oinf) -> int {
        const size_t k = inf.size();
        oxy.resize(8 * k); oinf.resize(k);
        return kh_mask_custom(srs, xy.data(), inf.data(), k, (const uint64_t*)blinders, k, oxy.data(), oinf.data());
    };
    // SRS::mask_custom in two halves around a commitment still running on the device: the blinding points [r_j] H first (host, ~5 us each), ...
    auto blinding_points = [&](const fe* blinders, size_t k, std::vector<uint64_t>& bxy, std::vector<uint8_t>& binf) -> int {
        std::vector<uint64_t> none(8 * k, 0); std::vector<uint8_t> at_inf(k, 1);
        bxy.resize(8 * k); binf.resize(k);
        return kh_mask_custom(srs, none.data(), at_inf.data(), k, (const uint64_t*)blinders, k, bxy.data(), binf.data());
    };
    // ... then one addition per commitment once its result is there
    auto mask_with = [&](const std::vector<uint64_t>& xy, const std::vector<uint8_t>& inf, const std::vector<uint64_t>& bxy, const std::vector<uint8_t>& binf,
                         std::vector<uint64_t>& oxy, std::vector<uint8_t>& oinf) -> int {
        const size_t k = inf.size();
        oxy.resize(8 * k); oinf.resize(k);
        return kh_points_add(curve, xy.data(), inf.data(), bxy.data(), binf.data(), k, oxy.data(), oinf.data());
    };
    auto scalar_challenge = [&](kh_sponge_t* sp, fe& o) -> int {
        uint64_t ch[2];
        int rc = kh_sponge_challenge(sp, ch); if (rc) return rc;
        return kh_scalar_challenge_to_field(curve, ch, o.l);
    };
    // ---- witness on the device: [w 0..14 | z] in evaluation form
    Dev ev; KP(ev.alloc(16 * NB));
    struct Tickets {                                  // un-waited MSM tickets: an error on the way out must not leave their pipeline slots taken
        uint64_t t[4] = {0, 0, 0, 0}; bool live[4] = {false, false, false, false};
        ~Tickets() { for (int i = 0; i < 4; i++) if (live[i]) { uint64_t xy[8 * COLUMNS]; uint8_t inf[COLUMNS]; (void)kh_msm_wait(t[i], xy, inf); } }
        int wait(int i, uint64_t* xy, uint8_t* inf) { live[i] = false; return kh_msm_wait(t[i], xy, inf); }
    } tickets;
    // The witness arrives over PCIe in 0.6 ms at 2^16 rows (31 MB), and nothing of the proof can start without it -- except the columns' own interpolation and
    // extension, which need one column each.  So the columns travel in groups (KH_PROVE_UPLOAD_GROUPS, default 3; 1 = one transfer as before) and every group but
    // the last has its copy -> iNTT -> LDE queued behind its transfer: that work (2/3 of 0.37 ms) runs underneath the next group's transfer instead of underneath
    // the commitment, the permutation aggregation finds an idle stream ~0.25 ms earlier.  The commitment stays ONE batched MSM behind the last group (submitting
    // it per group was measured in round 4: three submits cost more than the overlap gave back).
    Dev cf; KP(cf.alloc(16 * NB));                    // coefficient forms [w | z]
    Dev e8; KP(e8.alloc(16 * N8));
    const bool any_lib = (ix->live != 0) || nopt > 0;
    const size_t w8 = (!any_lib && !all_gates) ? PERMUTS : COLUMNS;     // generic + permutation read w0..w6 only
    // The witness extension (0.35 ms of throughput work) either right behind the interpolation -- it then runs underneath the transfer / the witness commitment, and the
    // small kernels of the permutation aggregation queue behind it -- or (KH_LDE_LATE=1) behind the aggregation, underneath the z commitment.
    static const bool lde_late = getenv("KH_LDE_LATE") && atoi(getenv("KH_LDE_LATE")) != 0;
    auto interpolate_extend = [&](size_t c0, size_t c1) -> int {       // columns c0 .. c1-1: evaluations -> coefficients -> d8
        if (c1 <= c0) return KH_OK;
        int rc_ = kh_dev_copy(cf.at(c0 * NB), ev.at(c0 * NB), (c1 - c0) * NB * 32); if (rc_) return rc_;
        rc_ = kh_ntt_dev(fid, cf.at(c0 * NB), logn, 1, c1 - c0); if (rc_) return rc_;
        const size_t e1 = c1 < w8 ? c1 : w8;
        if (!lde_late && e1 > c0) rc_ = kh_lde_dev(fid, cf.at(c0 * NB), logn, 3, e8.at(c0 * N8), e1 - c0);
        return rc_;
    };
    size_t cols_pending = 0;                           // first column whose interpolation / extension is not queued yet
    static const bool commit_per_group = getenv("KH_PROVE_COMMIT_PER_GROUP") && atoi(getenv("KH_PROVE_COMMIT_PER_GROUP")) != 0;
    size_t commit_groups = 0, group_c0[4] = {0, 0, 0, 0};
    if (witness) {
        KP_REQUIRE(rows + zk <= n, "NoRoomForZkInWitness: %zu rows + %zu zero-knowledge rows > %zu", rows, zk, n);
        if (rows + zk < n) KP(kh_dev_memset_zero(ev.p, 16 * NB * 32));
        const fe* z = draw(COLUMNS * zk);            // per column, from the LAST row backwards (prover.rs:254-266)
        std::vector<fe> zkr(COLUMNS * zk);
        for (size_t c = 0; c < COLUMNS; c++) for (size_t j = 0; j < zk; j++) zkr[c * zk + j] = z[c * zk + (zk - 1 - j)];
        KP(kh_dev_upload_2d(ev.at(n - zk), NB * 32, zkr.data(), zk * 32, zk * 32, COLUMNS));
        static const size_t groups_env = getenv("KH_PROVE_UPLOAD_GROUPS") ? (size_t)atoi(getenv("KH_PROVE_UPLOAD_GROUPS")) : 3;
        const size_t groups = (rows == 0 || groups_env < 1 || nch != 1 || n < 4096) ? 1 : (groups_env > COLUMNS ? COLUMNS : groups_env);
        for (size_t g = 0; g < groups && rows; g++) {
            const size_t c0 = COLUMNS * g / groups, c1 = COLUMNS * (g + 1) / groups;
            // (the destination rows are touched by nothing that is queued: no wait for the main stream, which holds the previous group's transforms)
            if (groups > 1) KP(kh_dev_upload_2d_unordered(ev.at(c0 * NB), NB * 32, witness + 4 * c0 * rows, rows * 32, rows * 32, c1 - c0));
            else KP(kh_dev_upload_2d(ev.p, NB * 32, witness, rows * 32, rows * 32, COLUMNS));
            if (g + 1 < groups) { KP(interpolate_extend(c0, c1)); cols_pending = c1; }
            // KH_PROVE_COMMIT_PER_GROUP=1 (experiment): the group's share of the witness commitment behind its transfer as well (a batch of its own)
            if (commit_per_group && groups > 1 && groups <= 3) {
                KP(kh_msm_submit(srs, (int)logn, 0, 0, ev.at(c0 * NB), n, c1 - c0, 1, &tickets.t[g])); tickets.live[g] = true; group_c0[g] = c0; group_c0[g + 1] = c1; commit_groups = g + 1;
            }
        }
    } else KP(kh_dev_copy(ev.p, witness_dev, COLUMNS * NB * 32));
    mark();
    SpongeH fq; KP(kh_sponge_new(KH_SPONGE_FQ, curve, &fq.s));
    KP(kh_sponge_absorb(fq.s, ix->digest.l, 1));
    {                                                 // prover.rs:276-279: the previous proofs' accumulator commitments
        size_t pos = 0;
        for (size_t j = 0; j < n_prev; j++) { KP(kh_sponge_absorb_g(fq.s, prev_comm_xy + 8 * pos, prev_comm_inf + pos, prev_comm_chunks[j])); pos += prev_comm_chunks[j]; }
    }
    Dev pub_c;
    std::vector<uint64_t> pub_xy; std::vector<uint8_t> pub_inf;
    if (ix->pub) {                                    // the negated public-input polynomial (prover.rs:281-309)
        KP_REQUIRE(ix->pub <= n, "more public inputs than rows");
        std::vector<fe> pe(n, zero);
        KP(kh_dev_download(pe.data(), ev.p, ix->pub * 32));
        for (size_t i = 0; i < ix->pub; i++) pe[i] = F.neg(pe[i]);
        KP(pub_c.alloc(NB)); KP(kh_dev_upload(pub_c.p, pe.data(), NB * 32));
        std::vector<uint64_t> cxy; std::vector<uint8_t> cinf;
        KP(commit_evals(pub_c.p, 1, cxy, cinf));
        std::vector<fe> ones(nch, one);
        KP(mask(cxy, cinf, ones.data(), pub_xy, pub_inf));
        KP(kh_ntt_dev(fid, pub_c.p, logn, 1, 1));
    } else { pub_xy = ix->zsel_xy; pub_inf = ix->zsel_inf; }
    KP(kh_sponge_absorb_g(fq.s, pub_xy.data(), pub_inf.data(), nch));
    pr->set_points(KH_PROOF_PUBLIC_COMM, pub_xy.data(), pub_inf.data(), nch);
    // ---- the proof's invariants (the accumulators end at 1, the divisions leave no remainder) are queued as device-side checks behind the steps that
    // produce them (kh_check_equal_dev: one bit each in a word on the device) and read ONCE, after the opening: a synchronous download per check
    // stalled the stream five times per proof, and the small host <-> device patches below (a one, two random rows, z_0 - 1) four times more.
    Dev chk; KP(chk.alloc(4));                         // [0]: the flags word; [1], [2]: the remainders of the two boundary divisions
    KP(kh_dev_memset_zero(chk.p, 4 * 32));
    uint32_t* const chk_flags = (uint32_t*)chk.p;
    enum { CHK_AGG = 0, CHK_Z = 1, CHK_REM = 2, CHK_BND = 3 };
    static const char* const chk_msg[4] = {"final value of the lookup aggregation is not 1 (lookup/constraints.rs:325-331)",
                                           "final value of the permutation accumulator is not 1 (permutation.rs:566-568)",
                                           "rest of division by vanishing polynomial (prover.rs:913-917): the witness does not satisfy the constraints",
                                           "permutation boundary division rest (permutation.rs:301-321)"};
    auto eager_check = [&](unsigned bit) -> int {      // KH_PROVE_EAGER_CHECK: the check queued just now, read back on the spot
        if (!eager) return KH_OK;
        uint32_t fl = 0;
        int rc_ = kh_dev_download(&fl, chk.p, 4); if (rc_) return rc_;
        if (fl & (1u << bit)) { kh::set_error("%s", chk_msg[bit]); return KH_E_INVALID; }
        return KH_OK;
    };
    auto set_const = [&](uint64_t* dst, const fe& val) { return kh_dev_fill_elements(dst, val.l, 1); };    // *dst = val, queued on the main stream
    // ---- witness commitments: one batched MSM per chunk of the Lagrange basis, queued before the columns are interpolated
    uint64_t& tk = tickets.t[3]; bool& have_tk = tickets.live[3];
    if (nch == 1 && !commit_groups) { KP(kh_msm_submit(srs, (int)logn, 0, 0, ev.p, n, COLUMNS, 1, &tk)); have_tk = true; }
    KP(interpolate_extend(cols_pending, COLUMNS));
    std::vector<uint64_t> wxy, wbx; std::vector<uint8_t> winf, wbi;
    const fe* w_blind = draw(COLUMNS * nch);          // blinder(num_chunks) per column, column by column (prover.rs:316-327)
    KP(blinding_points(w_blind, COLUMNS * nch, wbx, wbi));              // (underneath the MSM)
    if (commit_groups) {
        wxy.resize(8 * COLUMNS); winf.resize(COLUMNS);
        for (size_t g = 0; g < commit_groups; g++) KP(tickets.wait((int)g, &wxy[8 * group_c0[g]], &winf[group_c0[g]]));
    } else if (have_tk) { wxy.resize(8 * COLUMNS); winf.resize(COLUMNS); KP(tickets.wait(3, wxy.data(), winf.data())); }
    else KP(commit_evals(ev.p, COLUMNS, wxy, winf));
    std::vector<uint64_t> wcx; std::vector<uint8_t> wci;
    KP(mask_with(wxy, winf, wbx, wbi, wcx, wci));
    KP(kh_sponge_absorb_g(fq.s, wcx.data(), wci.data(), COLUMNS * nch));
    pr->set_points(KH_PROOF_W_COMM, wcx.data(), wci.data(), COLUMNS * nch);
    // ---- lookup argument, part 1 (prover.rs:383-633): joint combiner, combined table, sorted columns
    const kh_lookup_index* lk = ix->lk;
    const size_t ns = lk ? lk->mpr + 1 : 0, npat = lk ? lk->pats.size() : 0, lookup_rows = n - zk - 1;
    fe jc = zero, tic_t = zero, tic_c = zero;
    Dev d_table, d_sorted;
    const fe* s_blind = nullptr;
    Dev d_rt, d_rtc, rt8;                             // the proof's runtime contribution to the table's second column: d1, coefficients, d8
    const fe* rt_blind = nullptr;
    const bool has_rt = lk && lk->rtsel1;
    if (has_rt) {                                     // prover.rs:397-470: placed at the runtime rows, zero-knowledge rows drawn from the last row backwards, committed hiding
        std::vector<fe> rte(n, zero);
        memcpy(&rte[lk->rt_offset], runtime_values, n_runtime * 32);
        const fe* z = draw(zk);
        for (size_t j = 0; j < zk; j++) rte[n - zk + j] = z[zk - 1 - j];
        KP(d_rt.alloc(NB)); KP(d_rtc.alloc(NB));
        KP(kh_dev_upload(d_rt.p, rte.data(), NB * 32));
        KP(kh_dev_copy(d_rtc.p, d_rt.p, NB * 32));
        KP(kh_ntt_dev(fid, d_rtc.p, logn, 1, 1));
        std::vector<uint64_t> rxy, rcx; std::vector<uint8_t> rinf, rci;
        KP(commit_coeffs(d_rtc.p, n, nch, rxy, rinf));
        KP_REQUIRE(rinf.size() == nch, "unexpected chunk count of the runtime table");
        rt_blind = draw(nch);
        KP(mask(rxy, rinf, rt_blind, rcx, rci));
        KP(kh_sponge_absorb_g(fq.s, rcx.data(), rci.data(), nch));
        pr->set_points(KH_PROOF_LOOKUP_RUNTIME_COMM, rcx.data(), rci.data(), nch);
    }
    if (lk) {
        uint64_t chal[2] = {0, 0};
        if (lk->mjs > 1) KP(kh_sponge_challenge(fq.s, chal));       // joint_lookup_used (lookups.rs:90-110)
        KP(kh_scalar_challenge_to_field(curve, chal, jc.l));
        tic_c = fpow(F, jc, lk->mjs);                               // the constraints' table-id combiner (constraints.rs:424-440)
        tic_t = lk->tids ? tic_c : zero;                            // ... the table's, when the index has a table-id column (prover.rs:500-572)
        const size_t ntc = lk->tcols.size();
        {                                                           // the combined table: Horner over the table columns + tic * ids
            Prog p;
            const uint32_t c_rt = (uint32_t)(ntc + (lk->tids ? 1 : 0));               // the runtime column goes to the table's second column (prover.rs:455-464)
            auto col = [&](size_t k) { p.cell((uint32_t)k); if (k == 1 && has_rt) { p.cell(c_rt); p.add(); } };
            col(ntc - 1);
            for (size_t k = ntc - 1; k-- > 0;) { p.C(jc); p.mul(); col(k); p.add(); }
            std::vector<const uint64_t*> cols(lk->tcols);
            if (lk->tids) { p.C(tic_t); p.cell((uint32_t)ntc); p.mul(); p.add(); cols.push_back(lk->tids); }
            if (has_rt) cols.push_back(d_rt.p);
            std::vector<size_t> lens(cols.size(), n);
            KP(d_table.alloc(NB));
            KP(p.run(fid, cols, lens, n, 1, 1, 0, d_table.p));
        }
        std::vector<fe> vals(lk->mpr * n), table(n);
        {                                                           // the looked-up joint values, one column per lookup slot (0 = the dummy entry's value)
            Dev d_vals; KP(d_vals.alloc(lk->mpr * NB));
            LookupChallenges ch{}; ch.jc = jc; ch.tic = tic_t;
            std::vector<const uint64_t*> cols;
            for (size_t i = 0; i < COLUMNS; i++) cols.push_back(ev.at(i * NB));
            for (size_t k = 0; k < npat; k++) cols.push_back(lk->sel1[k]);
            std::vector<size_t> lens(cols.size(), n);
            for (size_t sl = 0; sl < lk->mpr; sl++) {
                Prog p; bool first = true;
                for (size_t k = 0; k < npat; k++) {
                    const Pattern& P = PATTERNS[lk->pats[k]];
                    if (sl >= (size_t)P.n) continue;
                    emit_joint(p, F, P.l[sl], ch); p.cell((uint32_t)(COLUMNS + k)); p.mul();
                    if (!first) p.add();
                    first = false;
                }
                KP(p.run(fid, cols, lens, n, 1, 1, 0, d_vals.at(sl * NB)));
            }
            KP(kh_dev_download(vals.data(), d_vals.p, lk->mpr * NB * 32));
            KP(kh_dev_download(table.data(), d_table.p, NB * 32));
        }
        std::vector<fe> srt(ns * (lookup_rows + 1)), full(ns * n, zero);
        size_t bad = 0;
        KP(kh_lookup_sorted((const uint64_t*)table.data(), lookup_rows, (const uint64_t*)vals.data(), n, lk->mpr, (uint64_t*)srt.data(), &bad));
        for (size_t k = 0; k < ns; k++) {                           // zk_patch (constraints.rs:35-48): the last zk_rows random, column by column
            memcpy(&full[k * n], &srt[k * (lookup_rows + 1)], (lookup_rows + 1) * 32);
            memcpy(&full[k * n + (n - zk)], draw(zk), zk * 32);
        }
        KP(d_sorted.alloc(ns * NB)); KP(kh_dev_upload(d_sorted.p, full.data(), ns * NB * 32));
        std::vector<uint64_t> sxy, scx; std::vector<uint8_t> sinf, sci;
        KP(commit_evals(d_sorted.p, ns, sxy, sinf));
        s_blind = draw(ns * nch);
        KP(mask(sxy, sinf, s_blind, scx, sci));
        KP(kh_sponge_absorb_g(fq.s, scx.data(), sci.data(), ns * nch));
        pr->set_points(KH_PROOF_LOOKUP_SORTED_COMM, scx.data(), sci.data(), ns * nch);
    }
    mark();
    fe beta, gamma;
    KP(kh_sponge_challenge_field(fq.s, beta.l)); KP(kh_sponge_challenge_field(fq.s, gamma.l));
    // ---- lookup argument, part 2 (prover.rs:635-673; constraints.rs:233-338): the aggregation, committed before z
    Dev d_agg;
    const fe* a_blind = nullptr;
    LookupChallenges lch{};
    if (lk) {
        const size_t mpr = lk->mpr;
        lch.jc = jc; lch.tic = tic_t; lch.beta = beta; lch.gamma = gamma; lch.gb1 = F.mul(gamma, F.add(one, beta));
        const fe b1m = fpow(F, F.add(one, beta), mpr);
        fe gp = one;
        for (size_t k = 0; k <= mpr; k++) { lch.prefactor[k] = F.mul(gp, b1m); gp = F.mul(gp, gamma); }     // the dummy entry's value is 0
        // columns: witness 0..14 | sorted 15..15+mpr | combined table | pattern selectors
        std::vector<const uint64_t*> cols;
        for (size_t i = 0; i < COLUMNS; i++) cols.push_back(ev.at(i * NB));
        for (size_t k = 0; k < ns; k++) cols.push_back(d_sorted.at(k * NB));
        const uint32_t c_table = (uint32_t)cols.size(); cols.push_back(d_table.p);
        const uint32_t c_sel0 = (uint32_t)cols.size();
        for (size_t k = 0; k < npat; k++) cols.push_back(lk->sel1[k]);
        std::vector<size_t> lens(cols.size(), n);
        Prog pn, pd;
        emit_numerator(pn, F, lk->pats, mpr, lch, c_sel0, c_table);
        emit_denominator(pd, mpr, lch, (uint32_t)COLUMNS);
        Dev num, den; KP(num.alloc(NB)); KP(den.alloc(NB)); KP(d_agg.alloc(NB));
        KP(kh_dev_memset_zero(num.p, NB * 32)); KP(kh_dev_memset_zero(den.p, NB * 32));
        KP(pn.run(fid, cols, lens, lookup_rows, 1, 1, 0, num.at(1)));
        KP(pd.run(fid, cols, lens, lookup_rows, 1, 1, 0, den.at(1)));
        KP(kh_batch_inversion_dev(fid, den.at(1), lookup_rows));
        KP(set_const(num.p, one)); KP(set_const(den.p, one));
        const uint32_t prod[6] = {KH_TOK_CELL, 0, KH_TOK_CELL, 2, KH_TOK_MUL, 0};
        const uint64_t* pc[2] = {num.p, den.p}; const size_t pl[2] = {n, n};
        KP(kh_expr_evaluations_dev(fid, prod, 3, pc, pl, 2, one.l, 1, n, 1, 1, 0, d_agg.p));
        KP(kh_field_scan_dev(fid, KH_SCAN_MUL, 0, d_agg.p, lookup_rows + 1));
        if (check) { KP(kh_check_equal_dev(d_agg.at(lookup_rows), 1, one.l, chk_flags, CHK_AGG)); KP(eager_check(CHK_AGG)); }      // before the random rows overwrite anything: row lookup_rows = n - zk - 1 is not one of them
        { const fe* rr = draw(zk); for (size_t j = 0; j < zk; j++) KP(set_const(d_agg.at(n - zk + j), rr[j])); }
        a_blind = draw(nch);
        std::vector<uint64_t> axy, acx; std::vector<uint8_t> ainf, aci;
        KP(commit_evals(d_agg.p, 1, axy, ainf));
        KP(mask(axy, ainf, a_blind, acx, aci));
        KP(kh_sponge_absorb_g(fq.s, acx.data(), aci.data(), nch));
        pr->set_points(KH_PROOF_LOOKUP_AGGREG_COMM, acx.data(), aci.data(), nch);
        KP(kh_sync());                                              // num / den are released at the end of this block
    }
    // ---- permutation aggregation z: numerators / denominators, batch inversion, running product (permutation.rs:510-568)
    fe bshift[7];
    for (int i = 0; i < 7; i++) bshift[i] = F.mul(beta, ix->shifts[i]);
    uint64_t* zcol = ev.at(COLUMNS * NB);
    Dev num, den; KP(num.alloc(NB)); KP(den.alloc(NB));
    {
        const uint64_t* cols[15]; size_t lens[15];
        for (size_t i = 0; i < PERMUTS; i++) { cols[i] = ev.at(i * NB); cols[PERMUTS + i] = ix->col1(COLUMNS + 2 + i); }
        cols[14] = ix->col1(COLUMNS + 1);
        for (int i = 0; i < 15; i++) lens[i] = n;
        fe consts[9]; consts[0] = gamma; consts[1] = beta; for (int i = 0; i < 7; i++) consts[2 + i] = bshift[i];
        std::vector<uint32_t> nt, dt;
        auto tok = [](std::vector<uint32_t>& v, uint32_t op, uint32_t a) { v.push_back(op); v.push_back(a); };
        for (uint32_t i = 0; i < 7; i++) {            // prod_i (w_i + sid beta shift_i + gamma), prod_i (w_i + sigma_i beta + gamma)
            tok(nt, KH_TOK_CELL, 2 * i); tok(nt, KH_TOK_CELL, 2 * 14); tok(nt, KH_TOK_CONST, 2 + i); tok(nt, KH_TOK_MUL, 0); tok(nt, KH_TOK_ADD, 0);
            tok(nt, KH_TOK_CONST, 0); tok(nt, KH_TOK_ADD, 0); if (i) tok(nt, KH_TOK_MUL, 0);
            tok(dt, KH_TOK_CELL, 2 * i); tok(dt, KH_TOK_CELL, 2 * (7 + i)); tok(dt, KH_TOK_CONST, 1); tok(dt, KH_TOK_MUL, 0); tok(dt, KH_TOK_ADD, 0);
            tok(dt, KH_TOK_CONST, 0); tok(dt, KH_TOK_ADD, 0); if (i) tok(dt, KH_TOK_MUL, 0);
        }
        // z_0 = 1, z_(i+1) = z_i num_i / den_i: the quotients of rows 0 .. n-2 go to z's rows 1 .. n-1 and the running product does the rest
        KP(kh_expr_evaluations_dev(fid, nt.data(), nt.size() / 2, cols, lens, 15, (const uint64_t*)consts, 9, n - 1, 1, 8, 0, num.p));
        KP(kh_expr_evaluations_dev(fid, dt.data(), dt.size() / 2, cols, lens, 15, (const uint64_t*)consts, 9, n - 1, 1, 8, 0, den.p));
        KP(kh_batch_inversion_dev(fid, den.p, n - 1));
        const uint32_t prod[6] = {KH_TOK_CELL, 0, KH_TOK_CELL, 2, KH_TOK_MUL, 0};
        const uint64_t* pc[2] = {num.p, den.p}; const size_t pl[2] = {n - 1, n - 1};
        KP(set_const(zcol, one));
        KP(kh_expr_evaluations_dev(fid, prod, 3, pc, pl, 2, one.l, 1, n - 1, 1, 8, 0, zcol + 4));
        KP(kh_field_scan_dev(fid, KH_SCAN_MUL, 0, zcol, n - zk + 1));
        if (check) { KP(kh_check_equal_dev(zcol + 4 * (n - zk), 1, one.l, chk_flags, CHK_Z)); KP(eager_check(CHK_Z)); }
        { const fe* rr = draw(2); KP(set_const(zcol + 4 * (n - zk + 1), rr[0])); KP(set_const(zcol + 4 * (n - zk + 2), rr[1])); }   // z's two random rows, in that order
        if (zk > 3) KP(kh_field_scan_dev(fid, KH_SCAN_MUL, 0, zcol + 4 * (n - zk + 2), zk - 2));
    }
    uint64_t* zc = cf.at(COLUMNS * NB);
    KP(kh_dev_copy(zc, zcol, NB * 32));
    KP(kh_ntt_dev(fid, zc, logn, 1, 1));
    if (nch == 1 && size == n) { KP(kh_msm_submit(srs, KH_BASIS_G, 0, 0, zc, n, 1, 1, &tk)); have_tk = true; }   // ... while z is extended to d8
    if (lde_late) KP(kh_lde_dev(fid, cf.p, logn, 3, e8.p, w8));
    KP(kh_lde_dev(fid, zc, logn, 3, e8.at(COLUMNS * N8), 1));
    std::vector<uint64_t> zxy, zbx; std::vector<uint8_t> zinf, zbi;
    const fe* z_blind = draw(nch);
    KP(blinding_points(z_blind, nch, zbx, zbi));
    if (have_tk) { zxy.resize(8); zinf.resize(1); KP(tickets.wait(3, zxy.data(), zinf.data())); }
    else KP(commit_coeffs(zc, n, nch, zxy, zinf));
    const size_t nzb = zinf.size();
    KP_REQUIRE(nzb == nch, "unexpected chunk count of z");
    std::vector<uint64_t> zcx; std::vector<uint8_t> zci;
    KP(mask_with(zxy, zinf, zbx, zbi, zcx, zci));
    KP(kh_sponge_absorb_g(fq.s, zcx.data(), zci.data(), nzb));
    pr->set_points(KH_PROOF_Z_COMM, zcx.data(), zci.data(), nzb);
    mark();
    fe alpha; KP(scalar_challenge(fq.s, alpha));
    fe alphas[3]; alphas[0] = fpow(F, alpha, ALPHA_PERM0); alphas[1] = F.mul(alphas[0], alpha); alphas[2] = F.mul(alphas[1], alpha);
    // ---- constraint rows on d8, quotient (prover.rs:794-917)
    Dev t8; KP(t8.alloc(N8));                         // the constraint rows on d8 (the reference keeps the generic gate's on d4: same polynomial, see below)
    {
        const uint64_t* cols[31];
        std::vector<uint64_t> consts(4 * 64);
        for (size_t i = 0; i < COLUMNS; i++) { cols[i] = e8.at(i * N8); cols[COLUMNS + i] = ix->col8(i); }
        cols[30] = ix->col8(COLUMNS);
        fe gp[2] = {one, alpha};
        const uint64_t* pc[31];
        for (size_t i = 0; i < COLUMNS; i++) pc[i] = e8.at(i * N8);
        for (size_t i = 0; i < PERMUTS; i++) pc[COLUMNS + i] = ix->col8(COLUMNS + 2 + i);
        pc[22] = e8.at(COLUMNS * N8); pc[23] = ix->col8(ix->ncol); pc[24] = ix->col8(ix->ncol + 1);
        for (int i = 25; i < 31; i++) pc[i] = pc[0];
        fe pp[10]; pp[0] = gamma; pp[1] = beta; pp[2] = alphas[0]; for (int i = 0; i < 7; i++) pp[3 + i] = bshift[i];
        KP(kh_gate_constants(fid, ix->gid_perm, nullptr, nullptr, (const uint64_t*)pp, 10, consts.data()));
        KP(kh_gate_evaluations_dev(fid, ix->gid_perm, pc, N8, consts.data(), (size_t)kh_gate_num_constants(ix->gid_perm), N8, 1, 8, 0, t8.p));
        // the double generic gate on ALL of d8, accumulated onto the permutation rows: the reference evaluates it on d4 and interpolates separately
        // (prover.rs:794-822, t4); its degree is below 4n, so the 8n-point interpolation of the sum gives the same polynomial -- the 4n-point iNTT and
        // the t4 buffer go away, and the kernel, memory-bound on whole cache lines either way, reads the same lines it read with stride 2
        KP(kh_gate_constants(fid, ix->gid_generic, nullptr, nullptr, (const uint64_t*)gp, 2, consts.data()));
        KP(kh_gate_evaluations_dev(fid, ix->gid_generic, cols, N8, consts.data(), (size_t)kh_gate_num_constants(ix->gid_generic), N8, 1, 8, 1, t8.p));
        for (size_t k = 0; k < 5 + nopt; k++) {      // the gate library on d8 (prover.rs:824-868): index(gate) * sum_i alpha^i constraint_i
            const bool live = k < 5 ? (((ix->live >> k) & 1u) || all_gates) : true;
            if (!live) continue;
            const int gid = k < 5 ? ix->lib_gate[k] : ix->optional[k - 5];
            const int nc = kh_gate_num_constants(gid);
            KP_REQUIRE(nc >= 0 && nc <= 64, "constants table of gate %d too large", gid);
            KP(kh_gate_constants(fid, gid, alpha.l, ix->endo.l, nullptr, 0, consts.data()));
            cols[30] = ix->col8(SEL0 + k);
            KP(kh_gate_evaluations_dev(fid, gid, cols, N8, consts.data(), (size_t)nc, N8, 1, 8, 1, t8.p));
        }
    }
    Dev lkc, lk8;                                      // coefficient forms / d8 of [sorted ... | aggregation | combined table]
    const size_t nl = lk ? ns + 2 : 0;
    if (lk) {                                         // the lookup constraints on d8 (prover.rs:874-903), powers alpha^24 ...
        const size_t mpr = lk->mpr;
        KP(lkc.alloc(nl * NB)); KP(lk8.alloc(nl * N8));
        KP(kh_dev_copy(lkc.p, d_sorted.p, ns * NB * 32));
        KP(kh_dev_copy(lkc.at(ns * NB), d_agg.p, NB * 32));
        KP(kh_dev_copy(lkc.at((ns + 1) * NB), d_table.p, NB * 32));
        KP(kh_ntt_dev(fid, lkc.p, logn, 1, nl));
        KP(kh_lde_dev(fid, lkc.p, logn, 3, lk8.p, nl));
        LookupChallenges cch = lch; cch.tic = tic_c;
        // columns: witness 0..14 | sorted | aggregation | table | pattern selectors | vanish, l0, lfinal (expr.rs:883-893)
        std::vector<const uint64_t*> cols;
        for (size_t i = 0; i < COLUMNS; i++) cols.push_back(e8.at(i * N8));
        for (size_t k = 0; k < nl; k++) cols.push_back(lk8.at(k * N8));
        const uint32_t c_sorted = (uint32_t)COLUMNS, c_agg = (uint32_t)(COLUMNS + ns), c_table = c_agg + 1, c_sel0 = c_table + 1;
        for (size_t k = 0; k < npat; k++) cols.push_back(lk->sel8[k]);
        const uint32_t c_vanish = (uint32_t)cols.size(), c_l0 = c_vanish + 1, c_lfinal = c_vanish + 2;
        for (int a = 0; a < 3; a++) cols.push_back(lk->atoms8[a]);
        std::vector<size_t> lens(cols.size(), N8);
        fe ap = fpow(F, alpha, ALPHA_PERM0 + 3);
        Prog p;
        // alpha^24 vanish (aggreg' denominator - aggreg numerator)
        p.C(ap); p.cell(c_vanish);
        p.cell(c_agg, 1); emit_denominator(p, mpr, cch, c_sorted); p.mul();
        p.cell(c_agg); emit_numerator(p, F, lk->pats, mpr, cch, c_sel0, c_table); p.mul();
        p.sub(); p.mul(); p.mul();
        // alpha^25 l0 (aggreg - 1), alpha^26 lfinal (aggreg - 1)
        for (int i = 0; i < 2; i++) { ap = F.mul(ap, alpha); p.C(ap); p.cell(i ? c_lfinal : c_l0); p.cell(c_agg); p.C(one); p.sub(); p.mul(); p.mul(); p.add(); }
        // the snake's shared elements: lfinal (s_i - s_i+1) for even i, l0 (...) for odd i
        for (size_t i = 0; i < mpr; i++) {
            ap = F.mul(ap, alpha);
            p.C(ap); p.cell((i & 1) ? c_l0 : c_lfinal); p.cell(c_sorted + (uint32_t)i); p.cell(c_sorted + (uint32_t)i + 1); p.sub(); p.mul(); p.mul(); p.add();
        }
        if (has_rt) {                                 // the constraints are padded to 3 + 4, then RT(x) selector_RT(x) (constraints.rs:658-680, runtime_tables.rs:59-66)
            KP(rt8.alloc(N8)); KP(kh_lde_dev(fid, d_rtc.p, logn, 3, rt8.p, 1));
            const uint32_t c_rt8 = (uint32_t)cols.size(); cols.push_back(rt8.p); cols.push_back(lk->rtsel8);
            lens.assign(cols.size(), N8);
            p.C(fpow(F, alpha, ALPHA_PERM0 + 3 + 7)); p.cell(c_rt8); p.cell(c_rt8 + 1); p.mul(); p.mul(); p.add();
        }
        KP(p.run(fid, cols, lens, N8, 1, 8, 1, t8.p));
    }
    KP(kh_ntt_dev(fid, t8.p, logn + 3, 1, 1));
    if (pub_c.p) {                                    // f = t + public (prover.rs:906-908)
        const uint64_t* ps[2] = {t8.p, pub_c.p}; const size_t ls[2] = {8 * n, n};
        fe sc[2] = {one, one};
        KP(kh_poly_lincomb_dev(fid, ps, ls, (const uint64_t*)sc, 2, t8.p, 8 * n));
    }
    Dev quot, rem; KP(quot.alloc(7 * NB)); KP(rem.alloc(NB));
    KP(kh_divide_by_vanishing_poly_dev(fid, t8.p, 8 * n, logn, quot.p, rem.p));
    if (check) { KP(kh_check_equal_dev(rem.p, n, nullptr, chk_flags, CHK_REM)); KP(eager_check(CHK_REM)); }
    Dev zm1, b1, b2; KP(zm1.alloc(NB)); KP(b1.alloc(NB)); KP(b2.alloc(NB));
    {
        const uint64_t* ps[1] = {zc}; const size_t ls[1] = {n};
        KP(kh_poly_lincomb_dev(fid, ps, ls, one.l, 1, zm1.p, n));
        {   // z_0 - 1, in place, on the stream
            const uint32_t prog[6] = {KH_TOK_CELL, 0, KH_TOK_CONST, 0, KH_TOK_SUB, 0};
            const uint64_t* c0[1] = {zm1.p}; const size_t l0[1] = {n};
            KP(kh_expr_evaluations_dev(fid, prog, 3, c0, l0, 1, one.l, 1, 1, 1, 0, 0, zm1.p));
        }
        KP(kh_dev_memset_zero(b1.p, NB * 32)); KP(kh_dev_memset_zero(b2.p, NB * 32));
        const fe pts2[2] = {one, fpow(F, ix->omega, n - zk)};
        uint64_t* dst[2] = {b1.p, b2.p};
        for (int i = 0; i < 2; i++) {                 // (z - 1) / (x - 1), (z - 1) / (x - omega^(n - zk)) (permutation.rs:301-321)
            KP(kh_divide_by_linear_async_dev(fid, zm1.p, n, pts2[i].l, dst[i], chk.at(1 + i)));
            if (check) { KP(kh_check_equal_dev(chk.at(1 + i), 1, nullptr, chk_flags, CHK_BND)); KP(eager_check(CHK_BND)); }
        }
        const uint64_t* qs[3] = {quot.p, b1.p, b2.p}; const size_t ql[3] = {7 * n, n - 1, n - 1};
        fe sc[3] = {one, alphas[1], alphas[2]};
        KP(kh_poly_lincomb_dev(fid, qs, ql, (const uint64_t*)sc, 3, quot.p, 7 * n));
    }
    std::vector<uint64_t> txy, tbx; std::vector<uint8_t> tinf, tbi;
    if (nch == 1 && size == n) { KP(kh_msm_submit(srs, KH_BASIS_G, 0, 0, quot.p, n, 7, 1, &tk)); have_tk = true; }   // the seven chunks as one batch
    const size_t ntb = 7 * nch;
    const fe* t_blind = draw(ntb);
    KP(blinding_points(t_blind, ntb, tbx, tbi));
    if (have_tk) { txy.resize(8 * 7); tinf.resize(7); KP(tickets.wait(3, txy.data(), tinf.data())); }
    else KP(commit_coeffs(quot.p, 7 * n, 7 * nch, txy, tinf));
    KP_REQUIRE(tinf.size() == ntb, "unexpected chunk count of t");
    std::vector<uint64_t> tcx; std::vector<uint8_t> tci;
    KP(mask_with(txy, tinf, tbx, tbi, tcx, tci));
    KP(kh_sponge_absorb_g(fq.s, tcx.data(), tci.data(), ntb));
    pr->set_points(KH_PROOF_T_COMM, tcx.data(), tci.data(), ntb);
    mark();
    fe zeta; KP(scalar_challenge(fq.s, zeta));
    const fe zetaw = F.mul(zeta, ix->omega);
    SpongeH fq_before; KP(kh_sponge_clone(fq.s, &fq_before.s));
    // ---- chunked evaluations at zeta, zeta omega (prover.rs:989-1004)
    std::vector<const uint64_t*> polys;
    polys.push_back(zc); polys.push_back(ix->colc(COLUMNS));
    for (size_t k = 0; k < 5; k++) polys.push_back(ix->colc(SEL0 + k));
    for (size_t i = 0; i < COLUMNS; i++) polys.push_back(cf.at(i * NB));
    for (size_t i = 0; i < COLUMNS; i++) polys.push_back(ix->colc(i));
    for (size_t i = 0; i + 1 < PERMUTS; i++) polys.push_back(ix->colc(COLUMNS + 2 + i));
    for (size_t k = 0; k < nopt; k++) polys.push_back(ix->colc(OPT0 + k));
    const size_t L0 = polys.size();                   // opening order of the lookup polynomials (prover.rs:1368-1420): sorted ..., aggregation, table, selectors
    for (size_t k = 0; k < nl; k++) polys.push_back(lkc.at(k * NB));
    const size_t nrt = has_rt ? 2 : 0;                // ... combined table, runtime table, runtime selector, pattern selectors
    if (has_rt) { polys.push_back(d_rtc.p); polys.push_back(lk->rtselc); }
    for (size_t k = 0; k < npat; k++) polys.push_back(lk->selc[k]);
    const size_t npoly = polys.size();
    const fe pts[2] = {zeta, zetaw};
    std::vector<fe> E(npoly * 2 * nch);               // polynomial j: E[(2 j + p) nch + c]
    // The same launch evaluates the chunks of sigma_6 and of the quotient t at both points: ft = perm_scalar sigma_6 - (zeta^n - 1) t is linear in
    // them, so ft(zeta), ft(zeta omega) -- the Fr-sponge absorbs the latter FIRST -- are known without a second launch + download behind the host's
    // computation of perm_scalar (round 5: ~0.1 ms of idle GPU between the two evaluation launches).
    std::vector<fe> E_ft(2 * 8 * nch);               // sigma_6: [p][c], c < nch; then t: [p][c], c < 7 nch
    std::vector<fe> pub_eval(2 * nch, zero);
    const uint64_t* const sig6_c = ix->colc(COLUMNS + 2 + PERMUTS - 1);
    {
        std::vector<const uint64_t*> ev(polys); ev.push_back(sig6_c); ev.push_back(quot.at(0));
        std::vector<size_t> lens(npoly, n), chs(npoly, nch);
        lens.push_back(n); chs.push_back(nch);
        lens.push_back(7 * n); chs.push_back(7 * nch);
        if (pub_c.p) { ev.push_back(pub_c.p); lens.push_back(n); chs.push_back(nch); }      // ... and the public-input polynomial's, when there is one
        std::vector<fe> all(E.size() + E_ft.size() + (pub_c.p ? 2 * nch : 0));
        KP(kh_evaluate_chunks_batch_dev(fid, ev.data(), lens.data(), chs.data(), ev.size(), size, (const uint64_t*)pts, 2, (uint64_t*)all.data()));
        std::copy(all.begin(), all.begin() + E.size(), E.begin());
        std::copy(all.begin() + E.size(), all.begin() + E.size() + E_ft.size(), E_ft.begin());
        if (pub_c.p) std::copy(all.begin() + E.size() + E_ft.size(), all.end(), pub_eval.begin());
    }
    // ---- ft = perm_scalar sigma_6 - (zeta^n - 1) t, chunk-linearised with zeta^max_poly_size (Maller; prover.rs:1147-1200)
    const fe zeta1 = fpow(F, zeta, n), zeta_srs = fpow(F, zeta, size), zetaw_srs = fpow(F, zetaw, size);
    auto comb = [&](size_t j, int p) { return horner(F, &E[(2 * j + p) * nch], nch, p ? zetaw_srs : zeta_srs); };
    const fe wz = fpow(F, ix->omega, n - zk);
    fe zkp = F.mul(F.mul(F.sub(zeta, wz), F.sub(zeta, F.mul(wz, ix->omega))), F.sub(zeta, fpow(F, ix->omega, n - 1)));
    fe scal = F.mul(F.mul(F.mul(comb(0, 1), beta), alphas[0]), zkp);
    for (size_t i = 0; i + 1 < PERMUTS; i++)          // w_i: polynomial 7 + i; sigma_i: polynomial 37 + i
        scal = F.mul(scal, F.add(F.add(gamma, F.mul(beta, comb(37 + i, 0))), comb(7 + i, 0)));
    scal = F.neg(scal);
    const fe m1 = F.neg(F.sub(zeta1, one));
    const size_t ft_len = size < 7 * n ? size : 7 * n;
    Dev ft; KP(ft.alloc(ft_len));
    {
        std::vector<const uint64_t*> segs; std::vector<size_t> lens; std::vector<fe> scs;
        const uint64_t* sig6 = sig6_c;
        fe pw = one;
        for (size_t c = 0; c < nch; c++) {            // f_chunked.linearize(zeta^srs_len)
            if (c * size < n) { const size_t ln = n - c * size < size ? n - c * size : size; segs.push_back(sig6 + 4 * c * size); lens.push_back(ln); scs.push_back(F.mul(scal, pw)); }
            pw = F.mul(pw, zeta_srs);
        }
        pw = one;
        for (size_t c = 0; c < 7 * nch; c++) {        // t_chunked.linearize(zeta^srs_len) * -(zeta^n - 1)
            if (c * size < 7 * n) { const size_t ln = 7 * n - c * size < size ? 7 * n - c * size : size; segs.push_back(quot.at(c * size)); lens.push_back(ln); scs.push_back(F.mul(m1, pw)); }
            pw = F.mul(pw, zeta_srs);
        }
        KP(kh_poly_lincomb_dev(fid, segs.data(), lens.data(), (const uint64_t*)scs.data(), segs.size(), ft.p, ft_len));
    }
    fe fte[2];                                        // ft at zeta, zeta omega from the chunk evaluations (the polynomial itself is only needed by the opening)
    for (int p = 0; p < 2; p++)
        fte[p] = F.add(F.mul(scal, horner(F, &E_ft[p * nch], nch, zeta_srs)), F.mul(m1, horner(F, &E_ft[2 * nch + p * 7 * nch], 7 * nch, zeta_srs)));
    const fe blinding_ft = F.mul(m1, horner(F, t_blind, ntb, zeta_srs));
    // ---- Fr-sponge: v, u (prover.rs:1206-1250, plonk_sponge.rs:92-155)
    fe v, u;
    {
        SpongeH fr, pd; KP(kh_sponge_new(KH_SPONGE_FR, curve, &fr.s)); KP(kh_sponge_new(KH_SPONGE_FR, curve, &pd.s));
        fe d; KP(kh_sponge_digest(fq.s, d.l)); KP(kh_sponge_absorb(fr.s, d.l, 1));
        {                                             // the digest of the previous challenges (prover.rs:1212-1219)
            size_t pos = 0;
            for (size_t j = 0; j < n_prev; j++) { KP(kh_sponge_absorb(pd.s, prev_chals + 4 * pos, prev_rounds[j])); pos += prev_rounds[j]; }
        }
        KP(kh_sponge_digest(pd.s, d.l)); KP(kh_sponge_absorb(fr.s, d.l, 1));
        std::vector<fe> flat; flat.reserve(1 + 2 * nch * (npoly + 1));
        flat.push_back(fte[1]);
        flat.insert(flat.end(), pub_eval.begin(), pub_eval.end());
        flat.insert(flat.end(), E.begin(), E.begin() + 2 * nch * L0);
        if (lk) {                                     // plonk_sponge.rs:92-155: aggregation, table, sorted ..., pattern selectors
            auto both = [&](size_t j) { flat.insert(flat.end(), E.begin() + 2 * nch * j, E.begin() + 2 * nch * (j + 1)); };
            both(L0 + ns); both(L0 + ns + 1);
            for (size_t k = 0; k < ns; k++) both(L0 + k);
            for (size_t k = 0; k < nrt + npat; k++) both(L0 + nl + k);
        }
        KP(kh_sponge_absorb(fr.s, (const uint64_t*)flat.data(), flat.size()));
        KP(scalar_challenge(fr.s, v)); KP(scalar_challenge(fr.s, u));
    }
    pr->set_elems(KH_PROOF_EVALS, E.data(), E.size());
    pr->set_elems(KH_PROOF_PUBLIC_EVALS, pub_eval.data(), pub_eval.size());
    pr->set_elems(KH_PROOF_FT_EVAL1, &fte[1], 1);
    mark();
    // ---- SRS::open on (public, ft, z, 6 selectors, w x 15, coefficients x 15, sigma x 6, optional selectors)
    size_t logs = 0; while (((size_t)1 << logs) < size) logs++;
    std::vector<uint64_t> lr_xy(16 * logs); std::vector<uint8_t> lr_inf(2 * logs);
    uint64_t delta[8], sg[8], z1[4], z2[4]; uint8_t dinf = 0, sginf = 0;
    {
        std::vector<const uint64_t*> op; std::vector<size_t> ol, oc;
        // the previous challenges' polynomials b_poly_coefficients(chals), non-hiding, opened first (prover.rs:1220-1262); their evaluations are
        // closed-form (RecursionChallenge::evals, proof.rs:455-494): one chunk, or two when the polynomial is twice the SRS size
        std::vector<Dev> prev_bufs(n_prev);
        std::vector<fe> prev_e0, prev_e1;             // chunk evaluations at zeta / zeta omega, polynomial after polynomial
        {
            size_t cpos = 0;
            for (size_t j = 0; j < n_prev; j++) {
                const unsigned k = prev_rounds[j];
                KP_REQUIRE(k <= 26, "previous challenge %zu has %u rounds", j, k);
                const size_t ln = (size_t)1 << k;
                const size_t want_chunks = ln <= size ? 1 : 2;
                KP_REQUIRE((ln == size || ln == 2 * size) && prev_comm_chunks[j] == want_chunks, "previous challenge %zu: 2^%u coefficients / %zu commitment chunks do not fit an SRS of %zu", j, k, prev_comm_chunks[j], size);
                std::vector<fe> bc(ln);
                KP(kh_b_poly_coefficients(fid, prev_chals + 4 * cpos, k, 1, (uint64_t*)bc.data()));
                KP(prev_bufs[j].alloc(ln)); KP(kh_dev_upload(prev_bufs[j].p, bc.data(), ln * 32));
                op.push_back(prev_bufs[j].p); ol.push_back(ln); oc.push_back(want_chunks);
                fe full[2];
                for (int p = 0; p < 2; p++) {         // b_poly(chals, x) = prod_i (1 + chals[i] x^(2^(k-1-i))) (commitment.rs:426-436)
                    std::vector<fe> pw(k ? k : 1); pw[0] = pts[p];
                    for (unsigned i = 1; i < k; i++) pw[i] = F.sqr(pw[i - 1]);
                    fe r = one;
                    for (unsigned i = 0; i < k; i++) r = F.mul(r, F.add(one, F.mul(load(prev_chals + 4 * (cpos + i)), pw[k - 1 - i])));
                    full[p] = r;
                }
                if (want_chunks == 1) { prev_e0.push_back(full[0]); prev_e1.push_back(full[1]); }
                else {
                    const fe d0 = horner(F, bc.data() + size, ln - size, zeta), d1 = horner(F, bc.data() + size, ln - size, zetaw);
                    prev_e0.push_back(F.sub(full[0], F.mul(d0, zeta_srs))); prev_e0.push_back(d0);
                    prev_e1.push_back(F.sub(full[1], F.mul(d1, zetaw_srs))); prev_e1.push_back(d1);
                }
                cpos += k;
            }
        }
        op.push_back(pub_c.p ? pub_c.p : ix->zero_poly); ol.push_back(pub_c.p ? n : 0); oc.push_back(nch);
        op.push_back(ft.p); ol.push_back(ft_len); oc.push_back(1);
        for (size_t j = 0; j < npoly; j++) { op.push_back(polys[j]); ol.push_back(n); oc.push_back(nch); }
        std::vector<fe> bl;                            // one blinder per chunk of every opened polynomial
        bl.insert(bl.end(), prev_e0.size(), zero);
        bl.insert(bl.end(), nch, one); bl.push_back(blinding_ft);
        bl.insert(bl.end(), z_blind, z_blind + nch);
        bl.insert(bl.end(), 6 * nch, one);
        bl.insert(bl.end(), w_blind, w_blind + COLUMNS * nch);
        bl.insert(bl.end(), (COLUMNS + PERMUTS - 1 + nopt) * nch, zero);
        if (lk) {                                     // sorted, aggregation, the combined table -- sum_i jc^i over its masked columns + the table-id combiner
            bl.insert(bl.end(), s_blind, s_blind + ns * nch);                       // (prover.rs:1384-1400) --, the non-hiding pattern selectors
            bl.insert(bl.end(), a_blind, a_blind + nch);
            fe tb = tic_t, pw = one;
            for (size_t i = 0; i < lk->tcols.size(); i++) { tb = F.add(tb, pw); pw = F.mul(pw, jc); }
            if (has_rt) {                              // the runtime column's blinders enter the combined table's through the joint combiner (prover.rs:1402-1415)
                for (size_t c = 0; c < nch; c++) bl.push_back(F.add(F.mul(jc, rt_blind[c]), tb));
                bl.insert(bl.end(), rt_blind, rt_blind + nch);
                bl.insert(bl.end(), nch, zero);
            } else bl.insert(bl.end(), nch, tb);
            bl.insert(bl.end(), npat * nch, zero);
        }
        Dev a_dev, b_dev; KP(a_dev.alloc(size)); KP(b_dev.alloc(size));
        size_t out_len = 0;
        KP(kh_combine_polys_dev(fid, op.data(), ol.data(), oc.data(), op.size(), v.l, size, a_dev.p, &out_len));
        KP(kh_b_init_dev(fid, (const uint64_t*)pts, 2, u.l, size, b_dev.p));
        // per chunk: combined_inner_product (commitment.rs:622-657) and the combined blinder
        fe blinding_factor = zero, cip = zero, ps = one;
        size_t bi = 0;
        auto take = [&](const fe& c0, const fe& c1) {
            blinding_factor = F.add(blinding_factor, F.mul(bl[bi++], ps));
            cip = F.add(cip, F.mul(ps, F.add(c0, F.mul(u, c1))));
            ps = F.mul(ps, v);
        };
        for (size_t c = 0; c < prev_e0.size(); c++) take(prev_e0[c], prev_e1[c]);
        for (size_t c = 0; c < nch; c++) take(pub_eval[c], pub_eval[nch + c]);
        take(fte[0], fte[1]);
        for (size_t j = 0; j < npoly; j++) for (size_t c = 0; c < nch; c++) take(E[(2 * j) * nch + c], E[(2 * j + 1) * nch + c]);
        KP_REQUIRE(bi == bl.size(), "blinders / evaluation chunks mismatch");
        const fe* ob = draw(2 * logs + 2);            // (rand_l, rand_r) per round, then d, r_delta
        KP(kh_ipa_open(srs, a_dev.p, size, b_dev.p, size, cip.l, blinding_factor.l, fq_before.s, (const uint64_t*)ob, 2 * logs + 2, lr_xy.data(), lr_inf.data(),
                       delta, &dinf, z1, z2, sg, &sginf));
    }
    KP_REQUIRE(rpos == need, "randomness count mismatch");
    if (check) {                                     // the deferred invariants, earliest first (the reference returns the first it meets)
        uint32_t fl = 0;
        KP(kh_dev_download(&fl, chk.p, 4));
        for (unsigned bit = 0; bit < 4; bit++) KP_REQUIRE(!(fl & (1u << bit)), "%s", chk_msg[bit]);
    }
    pr->set_points(KH_PROOF_LR, lr_xy.data(), lr_inf.data(), 2 * logs);
    pr->set_points(KH_PROOF_DELTA, delta, &dinf, 1);
    pr->set_points(KH_PROOF_SG, sg, &sginf, 1);
    fe zz[2] = {load(z1), load(z2)};
    pr->set_elems(KH_PROOF_Z1_Z2, zz, 2);
    fe ch[7] = {beta, gamma, alpha, zeta, v, u, jc};
    pr->set_elems(KH_PROOF_CHALLENGES, ch, lk ? 7 : 6);
    KP(kh_sync());                                   // every queued user of the buffers released below has finished
    mark();
    guard.p = nullptr;
    *out = pr;
    return KH_OK;
}

int kh_proof_section(const kh_proof_t* proof, int section, const uint64_t** limbs, const uint8_t** flags, size_t* count) {
    if (!proof || section < 0 || section > KH_PROOF_LOOKUP_RUNTIME_COMM || !limbs || !count) { kh::set_error("kh_proof_section: bad argument"); return KH_E_INVALID; }
    const kh_proof::Sec& s = proof->sec[section];
    *limbs = s.limbs.data(); *count = s.count;
    if (flags) *flags = s.points ? s.flags.data() : nullptr;
    return KH_OK;
}
int kh_proof_phase_seconds(const kh_proof_t* proof, double* seconds, size_t cap) {
    if (!proof || !seconds) { kh::set_error("kh_proof_phase_seconds: null argument"); return KH_E_INVALID; }
    for (size_t i = 0; i < cap && i < 6; i++) seconds[i] = proof->phase[i];
    return 6;
}
void kh_proof_free(kh_proof_t* proof) { delete proof; }

}  // extern "C"
