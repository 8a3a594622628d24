// msm.hpp -- internal interface between api.hip and msm.hip
#pragma once
#include "common.hpp"
#include "host_ec.hpp"

namespace kh {

struct MsmBasis {
    const void* pts = nullptr;       // device, n x 64 B affine x||y (Montgomery); with precomp: W tables of n points
    const uint8_t* inf = nullptr;    // device, nullable per-point infinity flags
    size_t n = 0;                    // points per table
    int precomp_c = 0;               // 0: plain basis; else window width of the precomputed tables
    size_t stride = 0;               // points between consecutive window tables (0: = n)
    size_t batch_stride = 0;         // >0: MSM j of a batch uses points [j*batch_stride, ...) (independent bases)
    const void* wide_pts = nullptr;  // a second table set of the SAME basis with wide windows (same stride), used for big single MSMs
    int wide_c = 0;
    bool glv = false;                // the tables hold 2^(c w) P for the lower 128 bits and phi(2^(c w) P) behind them: scalars are split k = k1 + k2 lambda (rebase.hip)
};

int msm_pick_window(size_t n);
// builds the window tables 2^(c*w) * P_i in place: `tables` holds W x n x 64 B with table 0 = the basis
int msm_precompute(Context& C, int curve, void* tables, const uint8_t* inf, size_t n, int c);
int msm_precompute_on(hipStream_t s, int curve, void* tables, size_t n, int c, void* scratch);      // asynchronous, scratch = (W - 1) x n x 128 bytes
// rebase.hip: the folded basis of the opening's late rounds, materialised from the c = 16 tables (see there)
int rebase_points(hipStream_t s, int curve, const uint64_t* coef, size_t Q, const void* tables, size_t stride, size_t N, void* B, void* part, void* lists,
                  hipEvent_t after_plan = nullptr);
const void* rebase_outputs(const void* part, size_t N);          // the N materialised points (XYZZ, 128 bytes each) inside `part`
int rebase_tables(hipStream_t s, int curve, const void* part, size_t N, const void* extra_affine_host, size_t extra, int c, void* scratch, void* tables, uint32_t* fail,
                  const uint64_t* glv_beta = nullptr);      // glv_beta (Montgomery, base field): build W/2 levels and phi of them
int msm_debug_glv_split(hipStream_t s, int field, const uint64_t* scalars_dev, size_t n, uint32_t* out_dev);
const uint32_t* msm_glv_lambda(int scalar_field);         // the eigenvalue the split's constants were generated for (canonical, 8 x 32-bit limbs)
size_t rebase_bucket_bytes(size_t N);
size_t rebase_part_bytes(size_t N);
size_t rebase_list_bytes(size_t Q);
static constexpr int MSM_PRECOMP_C = 16;          // window width of the precomputed tables
static constexpr int MSM_WIDE_C = 20;             // ... of the second table set big bases get: 13 windows, 2^19 buckets (msm.hip, "wide windows")
void msm_set_wide_min_n(size_t n);
void msm_set_sort_staging(unsigned entries, unsigned max_passes);   // k_part2_sort's staged scatter (kh_msm_set_sort_staging)
size_t msm_wide_min_n();                          // MSMs of at least this many points take the wide tables (KH_WIDE_MIN_N, default 2^19; 0 = never)
                                                  // measured (tools/wide_ab.py, pipelined Mscalar/s narrow -> wide): 2^17 537 -> 400, 2^18 655 -> 685, 2^19 746 -> 815,
                                                  // 2^20 896 -> 960..1000, 2^21 770 -> 976, 2^22 861 -> 992
static constexpr size_t MSM_PRECOMP_MIN_N = 1024; // smaller bases keep the plain per-window path
static constexpr int IPA_ROUND_C = 16;            // window width of the opening rounds' table set (KH_IPA_C overrides; < 16: a second, narrower set)
// enqueue all device work of k MSMs on slot S (returns immediately); msm_finish waits for it and does the host part
// use_graph: flags.  MSM_REPEATS: the caller repeats this exact MSM (same buffers and sizes): from the second call on the launch sequence is
// captured once into a hipGraph and replayed
static constexpr int MSM_REPEATS = 1, MSM_SPREAD_SCALARS = 2;      // (MSM_SPREAD_SCALARS: msm.hip, "the caller vouches ...")
static constexpr int MSM_LATENCY = 4;                              // the caller's critical path: the accumulation runs above the default wave priority too
// Host scalars of ONE MSM (k == 1) that are still on their way: msm_enqueue uploads them to scalars_dev itself, in `nev` chunks on the copy stream `cs`
// (pageable memory: the runtime stages each chunk while the previous one's k_digits already runs on the slot's stream), each chunk's digits launched
// behind its own event -- the digit pass of a 2^20 MSM hides under the upload, and with two MSMs in flight the whole upload hides under the other
// job's accumulation (kh_msm_submit_host).
struct MsmHostScalars { const uint64_t* host; hipStream_t cs; hipEvent_t* ev; int nev; };
int msm_enqueue(Context& C, MsmSlot& S, int curve, const MsmBasis& basis, size_t offset, const uint64_t* scalars_dev, size_t n, size_t k, int mont,
                int use_graph = 0, const MsmHostScalars* hs = nullptr);
// flag_seen: the caller has read the job's launch count from S.done_flag (the result is in S.pinned): no wait on the event
int msm_finish(Context& C, MsmSlot& S, uint64_t* out_xy, uint8_t* out_inf, bool flag_seen = false);
int debug_field_op(Context& C, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int debug_point_op(Context& C, int curve, int op, const uint64_t* p, const uint8_t* pinf, const uint64_t* q, const uint8_t* qinf, uint8_t* out, size_t n);

// ntt.hip
int ntt_build_twiddles(Context& C, int field, unsigned logn, int inverse, uint64_t* tab);
void ntt_trim(Context& C);
// ipa.hip
int ipa_fold_scalars(Context& C, int field, const uint64_t* lo, const uint64_t* hi, const uint64_t u[4], size_t n, uint64_t* out);
int ipa_inner_product(Context& C, int field, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]);
int ipa_fold_points_endo(Context& C, int curve, const uint64_t* g_lo, const uint64_t* g_hi, const uint64_t chal[2], size_t n, uint64_t* out_xy, uint8_t* out_inf);
int ipa_round_step(hipStream_t s, int field, int has_fold, const uint64_t* a, const uint64_t* b, const uint64_t* coef, size_t n, size_t cur2, size_t ncoef,
                   const uint64_t u[4], const uint64_t uinv[4], uint64_t* a2, uint64_t* b2, uint64_t* coef2,
                   const uint64_t rand_l[4], const uint64_t rand_r[4], uint64_t* sc, uint64_t* partial, unsigned* counter);
int ipa_round_prepare(hipStream_t s, int field, const uint64_t* a, const uint64_t* b, const uint64_t* coef, size_t n, size_t Nj,
                      const uint64_t rand_l[4], const uint64_t rand_r[4], uint64_t* sc, uint64_t* partial);
int ipa_round_fold(hipStream_t s, int field, const uint64_t* a, const uint64_t* b, const uint64_t* coef, size_t Nj, size_t ncoef,
                   const uint64_t u[4], const uint64_t uinv[4], uint64_t* a2, uint64_t* b2, uint64_t* coef2);
int ipa_sg_split(hipStream_t s, int field, const uint64_t* coef, size_t n, int has_fold, const uint64_t u[4], uint64_t* out);
int bpoly_run(hipStream_t s, int field, const uint64_t* chals_dev, unsigned rounds, size_t k, const uint64_t* rs_dev, uint64_t* out_dev);
// poly.hip
int poly_lincomb(Context& C, int field, const uint64_t* const* segs_dev, const size_t* lens, const uint64_t* scales, size_t m, uint64_t* out_dev, size_t out_len);
int poly_b_init(Context& C, int field, const uint64_t* elm, const uint64_t* scales, size_t k, size_t n, uint64_t* out_dev);
int poly_eval_chunks(Context& C, int field, const uint64_t* const* polys_dev, const size_t* lens, const size_t* num_chunks, size_t m, size_t chunk,
                     const uint64_t* points, size_t npts, uint64_t* out);
int poly_div_vanishing(Context& C, int field, const uint64_t* f_dev, size_t len, size_t n, uint64_t* q_dev, uint64_t* r_dev);
int poly_coset_ntt(Context& C, int field, const uint64_t* coeffs_dev, unsigned log2_n, const uint64_t shift[4], uint64_t* out_dev, size_t batch);
int poly_scan(Context& C, int field, int op, int rev, uint64_t* data_dev, size_t n);
int poly_batch_inversion(Context& C, int field, uint64_t* v_dev, size_t n);
int poly_divide_by_linear(Context& C, int field, const uint64_t* f_dev, size_t len, const uint64_t a[4], uint64_t* q_dev, uint64_t rem[4], uint64_t* rem_dev = nullptr);
int poly_check_equal(Context& C, const uint64_t* v_dev, size_t n, const uint64_t* expect, uint32_t* flags_dev, unsigned bit);
// expr.hip
// the gate library as compiled kernels (gates.hip; generated from the same expression DAGs as the token programs)
int gate_count();
const char* gate_name(int gate);
int gate_num_constants(int gate);
int gate_constants(int field, int gate, const uint64_t* alpha, const uint64_t* endo, const uint64_t* params, size_t nparams, uint64_t* out);
int gate_run(Context& C, int field, int gate, const uint64_t* const* cols_dev, size_t len, const uint64_t* consts, size_t nconsts, size_t rows,
             unsigned stride, unsigned next_shift, int accumulate, uint64_t* out_dev);
int expr_run(Context& C, int field, const uint32_t* prog, size_t ntok, const uint64_t* const* cols_dev, const size_t* col_len, size_t ncols,
             const uint64_t* consts, size_t nconsts, size_t rows, unsigned stride, unsigned next_shift, int accumulate, uint64_t* out_dev);
// host_srs.cpp
void scalar_challenge_to_field(int field, const uint64_t chal[2], const uint64_t endo[4], uint64_t out[4]);
void host_window_multiples(int curve, const uint64_t xy[8], int W, int c, uint64_t* out_xy);   // out[w] = 2^(c w) P, affine
void host_field_inverse(int field, const uint64_t a[4], uint64_t out[4]);
void endo_coefficient(int field, uint64_t out[4]);
void curve_endos(int curve, uint64_t endo_q[4], uint64_t endo_r[4]);
int ipa_fold_points(Context& C, int curve, const uint64_t* g_lo, const uint64_t* g_hi, const uint64_t u[4], size_t n, uint64_t* out_xy, uint8_t* out_inf);
// srs_gen.hip
int srs_generate_device(Context& C, int curve, size_t start, size_t count, void* out_xy_dev);
// lagrange.hip
int lagrange_run(Context& C, int curve, const void* g_dev, size_t srs_size, unsigned log_n, unsigned chunk, void* out_xy_dev, uint8_t* out_inf_dev);
khost::fe ntt_host_root(int field, unsigned logn, int inverse);   // omega_{2^logn} (or its inverse), Montgomery
unsigned ntt_max_logr();
int ntt_set_max_logr(unsigned v);            // 4..10, 0 = default (kh_ntt_set_max_logr)
int ntt_run(Context& C, int field, uint64_t* data_dev, unsigned log2_n, int inverse, size_t batch);
int lde_run(Context& C, int field, const uint64_t* coeffs_dev, unsigned log2_n, unsigned log2_blowup, uint64_t* out_dev, size_t batch);

}  // namespace kh
