/*
 * kimchi_hip.h -- C ABI of the MI355X-native MSM + NTT hot path for Kimchi.
 *
 * This is the drop-in boundary: a Rust shim crate implements
 * poly-commitment's `trait SRS<G>` / `trait OpenProof<G>` by delegating the MSMs to
 * the kh_msm* entry points, and kimchi's evaluation-domain transforms by delegating
 * to kh_ntt / kh_lde.  Plain C, no exceptions, no torch types.  Every function
 * returns 0 on success or a negative KH_E_* code; kh_last_error() describes the last
 * failure on the calling thread.  All functions are thread-safe (the reference calls
 * SRS::commit_evaluations_non_hiding from 15 rayon workers at once,
 * kimchi/src/prover.rs:329-351).
 *
 * Wire format (zero-copy from Rust):
 *   field element = 4 x uint64_t little-endian limbs in Montgomery form, R = 2^256,
 *                   i.e. ark-ff's in-memory Fp256<MontBackend<_,4>> (the reference
 *                   itself reinterprets Fp as [u64;4]: kimchi/src/cached_prover_index.rs:502-539);
 *   affine point  = x || y = 8 limbs (64 bytes), infinity passed out of band
 *                   (ark-ec Affine{x,y,infinity} has no stable layout: the shim packs it);
 *   curve id      : 0 = Vesta  (coordinates in Fq, scalars in Fp)
 *                   1 = Pallas (coordinates in Fp, scalars in Fq)
 *   field id      : 0 = Fp, 1 = Fq.
 * Results are returned affine because the Fiat-Shamir sponge absorbs affine
 * coordinates (poly-commitment/src/commitment.rs:503-514).
 */
#ifndef KIMCHI_HIP_H
#define KIMCHI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KH_OK 0
#define KH_E_INVALID (-1)    /* bad argument (null pointer, size, id) */
#define KH_E_DEVICE (-2)     /* HIP runtime error / no device */
#define KH_E_NOTFOUND (-3)   /* basis not registered on this SRS */
#define KH_E_NOMEM (-4)

#define KH_CURVE_VESTA 0
#define KH_CURVE_PALLAS 1
#define KH_FIELD_FP 0
#define KH_FIELD_FQ 1
#define KH_BASIS_G (-1)      /* the monomial basis g of the SRS */

typedef struct kh_srs kh_srs_t;
typedef struct kh_sponge kh_sponge_t;      /* host-side Fiat-Shamir sponge, see the end of this file */

/* ---- device ------------------------------------------------------------- */
int kh_device_count(void);
/* Initialises a device (streams, workspaces) and makes it the calling thread's current device; the FIRST device
 * initialised is also the process default, i.e. what every other thread uses until it chooses otherwise.  Idempotent;
 * kh_init(-1) picks the current choice, else LOCAL_RANK, else device 0.  One process per GPU (torch.distributed
 * ranks) needs nothing else.  A single process may also drive SEVERAL GPUs: each device has its own context (streams,
 * four MSM pipeline slots, caches); an SRS handle lives on the device that was current when it was created and every
 * entry point taking a handle (kh_msm*, kh_commit*, kh_ipa_*, ...) runs on the handle's device whatever the calling
 * thread's current device is -- a Rust prover holds Vesta on GPU 0 and Pallas on GPU 1 (BASELINE config 5), or shards
 * one MSM over 8 handles created with kh_srs_create_device_range (config 4), from ordinary rayon threads.  Entry points
 * WITHOUT a handle (kh_ntt*, kh_lde*, kh_dev_alloc, the vector steps) use the calling thread's current device. */
int kh_init(int device_id);
int kh_set_device(int device_id);     /* thread-local: this thread's current device from now on (initialised on first use) */
int kh_get_device(void);              /* the calling thread's current device (-1 before any initialisation) */
/* Several prover threads in one process: between _begin and _end the calling thread works on a library context of its own on its current device -- own
 * main stream, MSM pipeline slots, workspaces and lock -- instead of the device's shared one, so that independent provers neither queue their vector steps
 * on one stream nor serialise their launches on one mutex (measured: four provers 150 -> 165-176 proofs/s; four processes: 177).  Everything the thread
 * queues in between is complete when _end returns; tickets of kh_msm_submit must be waited for before _end; work the thread queued on the shared context
 * before _begin is ordered in front.  SRS handles, device buffers and indexes are process-wide as before (a handle still runs one opening at a time).
 * kh_prove* do this themselves.  Contexts are pooled per device and reused. */
int kh_private_context_begin(void);
int kh_private_context_end(void);
int kh_private_context_active(void);   /* 1 if the calling thread is between _begin and _end on its current device */
/* per-phase HIP events behind kh_last_timings on the calling thread's current context (the private one between _begin and _end): OFF by default -- an event
 * between two kernels costs the stream 6-10 us of idle time; tools that read kh_last_timings switch them on */
int kh_set_phase_timers(int on);
int kh_trim(void);                    /* current device: free cached twiddle tables, scratch and idle MSM workspaces (rebuilt on demand) */
const char *kh_last_error(void);

/* ---- SRS: device-resident bases ------------------------------------------
 * Replaces the `g: Vec<G>` half of `ipa::SRS<G>` (poly-commitment/src/ipa.rs:53-75):
 * uploads the n monomial-basis points once; Rust keeps owning its own copy. */
int kh_srs_create(int curve, const uint64_t *g_xy /* n x 8 limbs */, size_t n, kh_srs_t **out);
/* Tuning knob (process-wide): bases of at least n points get a SECOND window-table set with 20-bit windows when they are created (13 instead of 16
 * table additions per scalar, 2^19 buckets, +13/16 of the table memory), and single MSMs of at least n scalars over them take it.  Default 2^19
 * (KH_WIDE_MIN_N), 0 = never.  Results are the same group elements either way; the tests lower it to run the wide path at small sizes. */
int kh_msm_set_wide_min_n(size_t n);
/* Tuning knob (process-wide) of the wide path's sort: the second sorting pass collects a partition's entries in LDS and writes them out as coalesced runs, in
 * at most `max_passes` passes of `entries` entries each (default 28,672 and 2: KH_PART2_STAGE / KH_PART2_MAXPASS); a partition that would need more passes, or
 * entries = 0, takes the direct scatter.  Same results under every setting; the tests shrink it to run the multi-pass and the fallback branches at small sizes. */
int kh_msm_set_sort_staging(unsigned entries, unsigned max_passes);
/* The wide set is OPTIONAL: if it does not fit in device memory when the handle is created, the handle is created without it (the narrow tables serve every
 * MSM, ~8 % slower at 2^20) and no error is reported.  kh_srs_set_wide_tables(srs, 0) gives an existing wide set back (832 MiB at 2^20 points, 3.3 GiB at
 * 2^22) -- no MSM over the handle may be in flight, as for kh_srs_free --, (srs, 1) builds it now whatever the threshold says and fails with KH_E_NOMEM if it
 * does not fit; kh_srs_has_wide_tables tells which state the handle is in. */
int kh_srs_set_wide_tables(kh_srs_t *srs, int on);
int kh_srs_has_wide_tables(const kh_srs_t *srs);
void kh_srs_free(kh_srs_t *srs);
size_t kh_srs_size(const kh_srs_t *srs);
int kh_srs_curve(const kh_srs_t *srs);    /* KH_CURVE_VESTA / KH_CURVE_PALLAS */
int kh_srs_device(const kh_srs_t *srs);   /* the device the handle's tables live on */

/* SRS::create (poly-commitment/src/ipa.rs:751-778) on the host: g_start .. g_{start+count-1}
 * (Blake2b-512 of the big-endian u32 index -> Shallue-van de Woestijne map, groupmap/src/lib.rs:74-189)
 * and the blinding base h (ipa.rs:765-772).  Affine x||y, Montgomery.  threads <= 0: all host cores. */
int kh_srs_generate(int curve, size_t start, size_t count, uint64_t *out_xy, int threads);
int kh_srs_h(int curve, uint64_t out_xy[8]);
/* GroupMap::to_group (groupmap/src/lib.rs:167-189): the U base of SRS::open / verify, u_base = to_group(sponge.challenge_fq())
 * (ipa.rs:909-913).  t: a base-field element, Montgomery limbs.  Host code. */
int kh_group_map_to_group(int curve, const uint64_t t[4], uint64_t out_xy[8]);

/* SRS::create(depth) entirely on the device (csrc/srs_gen.hip: Blake2b + group map + Tonelli-Shanks per
 * thread; ~15 ms for 2^20 points), already expanded to the MSM window tables; kh_srs_get_g reads
 * g[offset .. offset+count) back for the caller's own `g: Vec<G>`. */
int kh_srs_create_device(int curve, size_t depth, kh_srs_t **out);
/* the slice g[start .. start+count) of a larger SRS (one rank of a point-range-sharded MSM) */
int kh_srs_create_device_range(int curve, size_t start, size_t count, kh_srs_t **out);
int kh_srs_get_g(kh_srs_t *srs, size_t offset, size_t count, uint64_t *out_xy);

/* Registers chunk `chunk` of the Lagrange basis for the domain of size 2^log2_domain
 * (`SRS::get_lagrange_basis`, ipa.rs:780-801; entries computed by ipa.rs:1065-1172):
 * n = 2^log2_domain points, `inf` nullable per-point infinity flags. */
int kh_srs_set_lagrange(kh_srs_t *srs, unsigned log2_domain, unsigned chunk,
                        const uint64_t *xy, const uint8_t *inf, size_t n);
/* Computes the same basis on the device from g (group iNTT + batch normalisation,
 * ipa.rs:1065-1172) and registers every chunk.  Optional read-back via
 * kh_srs_get_lagrange (out_xy: n x 8 limbs, out_inf: n bytes). */
int kh_srs_compute_lagrange(kh_srs_t *srs, unsigned log2_domain);
int kh_srs_get_lagrange(kh_srs_t *srs, unsigned log2_domain, unsigned chunk,
                        uint64_t *out_xy, uint8_t *out_inf);
int kh_srs_lagrange_chunks(const kh_srs_t *srs, unsigned log2_domain);   /* 0 if not registered */

/* ---- MSM ------------------------------------------------------------------
 * Replaces ark_ec::VariableBaseMSM::{msm, msm_bigint} at the call sites
 * poly-commitment/src/ipa.rs:649,658-659,672 (commit_non_hiding) and
 * commitment.rs:382 (PolyComm::multi_scalar_mul, reached from
 * commit_evaluations_non_hiding ipa.rs:706-728).
 *
 * basis = KH_BASIS_G: out = sum_{i<n} scalars[i] * g[offset + i]
 * basis = log2_domain: out = sum_{i<n} scalars[i] * L_{offset+i}[chunk]
 * scalars: n x 4 limbs; scalars_are_montgomery selects Fp-in-memory (1) vs
 * into_bigint() canonical form (0, what msm_bigint receives).  Like msm_bigint,
 * uses min(n, basis_len - offset) pairs.  Result affine + infinity flag. */
int kh_msm(kh_srs_t *srs, int basis, unsigned chunk, size_t offset,
           const uint64_t *scalars, size_t n, int scalars_are_montgomery,
           uint64_t out_xy[8], uint8_t *out_is_inf);

/* k MSMs over the same basis window [offset, offset+n), scalars = k x n x 4 limbs
 * (the 15 witness-column commits of prover.rs:329-351, the 7 chunk commits of
 * ipa.rs:663-676 with per-chunk offsets handled by the host mirror). */
int kh_msm_batch(kh_srs_t *srs, int basis, unsigned chunk, size_t offset,
                 const uint64_t *scalars, size_t n, size_t k, int scalars_are_montgomery,
                 uint64_t *out_xy /* k x 8 */, uint8_t *out_is_inf /* k */);

/* Ad-hoc bases (IPA rounds ipa.rs:943-961, batch verifier ipa.rs:474-499). */
int kh_msm_points(int curve, const uint64_t *xy /* n x 8 */, const uint8_t *inf /* nullable */,
                  const uint64_t *scalars, size_t n, int scalars_are_montgomery,
                  uint64_t out_xy[8], uint8_t *out_is_inf);

/* k independent MSMs of the same length n with their own bases (xy: k x n x 8, inf: k x n or NULL,
 * scalars: k x n x 4) in one pass -- the L and R commitments of an IPA round (ipa.rs:943-961) are
 * two such MSMs that do not depend on each other. */
int kh_msm_points_batch(int curve, const uint64_t *xy, const uint8_t *inf, const uint64_t *scalars,
                        size_t n, size_t k, int scalars_are_montgomery,
                        uint64_t *out_xy /* k x 8 */, uint8_t *out_is_inf /* k */);

/* Sum of n affine points on the host (a handful of group additions): the fold of the per-GPU partial
 * sums of a point-range-sharded MSM after the all-gather, or `(r1 + r2).into_affine()` of ipa.rs:661. */
int kh_points_sum(int curve, const uint64_t *xy, const uint8_t *inf, size_t n, uint64_t out_xy[8], uint8_t *out_is_inf);
/* out_j = a_j + b_j for n pairs of affine points on the host (one field inversion in all; NULL flags = no point at infinity).  With
 * kh_mask_custom over commitments at infinity -- the blinding points [r_j] H, which do not depend on the commitment -- this is SRS::mask_custom
 * (ipa.rs:605-622) in two halves: the scalar multiplications run while the device computes the commitment, only n additions follow its result. */
int kh_points_add(int curve, const uint64_t *a_xy, const uint8_t *a_inf, const uint64_t *b_xy, const uint8_t *b_inf, size_t n,
                  uint64_t *out_xy /* n x 8 */, uint8_t *out_is_inf /* n */);

/* One MSM over a basis sharded by POINT RANGE over R handles (BASELINE config 4): shard r holds g[o_r, o_r + n_r), o_r =
 * n_0 + ... + n_(r-1), n_r = kh_srs_size(shards[r]), on whatever device it was created on (kh_srs_create_device_range after
 * kh_set_device: one shard per GPU, or several per GPU).  Each shard reduces its slice of the scalars with the full single-GPU
 * pipeline on its own device, concurrently; the R partial sums are folded on the host (kh_points_sum: RCCL has no group-addition
 * reduction, and R x 64 bytes do not need one).  `scalars`: host, N x 4 limbs, N = sum n_r (fewer: the tail shards get the rest / nothing).
 * This is the whole of config 4 for a one-process caller (the Rust shim); one-process-per-GPU deployments all-gather the partials
 * instead (proof_systems_amd/sharded.py over torch.distributed / RCCL). */
int kh_msm_sharded(kh_srs_t *const *shards, size_t R, const uint64_t *scalars, size_t n, int scalars_are_montgomery,
                   uint64_t out_xy[8], uint8_t *out_is_inf);
/* The same with the scalars already resident: scalars_dev[r] = shard r's slice (counts[r] <= n_r elements) on shard r's device.
 * All R jobs are submitted before the first is waited for. */
int kh_msm_sharded_dev(kh_srs_t *const *shards, size_t R, const uint64_t *const *scalars_dev, const size_t *counts,
                       int scalars_are_montgomery, uint64_t out_xy[8], uint8_t *out_is_inf);

/* ---- IPA round vector operations (poly-commitment/src/ipa.rs:980-1006) -----
 * out[i] = lo[i] + u * hi[i]      (a' = a_lo + u^-1 a_hi with u := u^-1; b' = b_lo + u b_hi)          */
int kh_ipa_fold_scalars(int field, const uint64_t *lo, const uint64_t *hi, const uint64_t u[4], size_t n, uint64_t *out);
/* <a, b> = sum a_i b_i (utils/src/field_helpers.rs:273-279) */
int kh_inner_product(int field, const uint64_t *a, const uint64_t *b, size_t n, uint64_t out[4]);
/* g'[i] = g_lo[i] + [u] g_hi[i], affine: CommitmentCurve::combine_one (commitment.rs:576-579); the
 * endo variant used by the prover (combine_one_endo, ipa.rs:1006) yields the same group elements for
 * u = u_pre.to_field(endo_r).  u: Montgomery limbs of the scalar field. */
int kh_ipa_fold_points(int curve, const uint64_t *g_lo_xy, const uint64_t *g_hi_xy, const uint64_t u[4], size_t n,
                       uint64_t *out_xy, uint8_t *out_inf);
/* CommitmentCurve::combine_one_endo (commitment.rs:581-589 -> combine.rs:292-340; prover call site
 * ipa.rs:1006): g'[i] = g_lo[i] + [chal.to_field(endo_r)] g_hi[i] by the Halo endo ladder -- 64 x
 * (doubling + mixed addition with +-g_hi or +-phi(g_hi)) instead of a 255-bit double-and-add.
 * chal: the 128-bit ScalarChallenge prechallenge as two canonical (non-Montgomery) LE limbs. */
int kh_ipa_fold_points_endo(int curve, const uint64_t *g_lo_xy, const uint64_t *g_hi_xy, const uint64_t chal[2], size_t n,
                            uint64_t *out_xy, uint8_t *out_inf);
/* endos::<G>() (ipa.rs:214-231): endo_q in the base field, endo_r in the scalar field (Montgomery limbs)
 * with phi(P) = (endo_q x, y) = [endo_r] P.  Host-only, needs no device. */
int kh_endos(int curve, uint64_t endo_q[4], uint64_t endo_r[4]);
/* ScalarChallenge::to_field(&endo_r) (poseidon/src/sponge.rs:190-226) for the curve's scalar field:
 * chal = the 128-bit prechallenge (two canonical LE limbs), out = Montgomery limbs.  Host-only. */
int kh_scalar_challenge_to_field(int curve, const uint64_t chal[2], uint64_t out[4]);

/* ---- coefficient vectors between the transforms / commitments and the opening, device-resident ----
 * kh_combine_polys_dev = combine_polys for polynomials in coefficient form (poly-commitment/src/utils.rs:103-206):
 *   polynomial i (device pointer, lens[i] coefficients) is cut into num_chunks[i] chunks of srs_length (the length
 *   of its blinder PolyComm); chunk number t overall is scaled by polyscale^t; out_dev receives srs_length
 *   coefficients (zero above *out_len = the longest chunk), ready for kh_ipa_begin_dev.  The combined blinder
 *   (a sum of m scalars) stays with the caller.
 * kh_b_init_dev: b[j] = sum_i evalscale^i elm_i^j, j < padded_len (poly-commitment/src/ipa.rs:863-888).
 * kh_evaluate_chunks_dev: to_chunked_polynomial(num_chunks, chunk_size).evaluate_chunks(x) for each of npts points
 *   (utils/src/dense_polynomial.rs:50-69, chunked_polynomial.rs:21-28; the zeta / zeta*omega evaluations of
 *   kimchi/src/prover.rs:1000-1170): out[p][c] = chunk_c(points[p]), npts x num_chunks x 4 limbs on the host.
 * kh_divide_by_vanishing_poly_dev: f = q (x^n - 1) + r (kimchi/src/prover.rs:903): q_dev gets len - n coefficients
 *   (none if len <= n), r_dev n coefficients. */
int kh_combine_polys_dev(int field, const uint64_t *const *polys_dev, const size_t *lens, const size_t *num_chunks, size_t m,
                         const uint64_t polyscale[4], size_t srs_length, uint64_t *out_dev, size_t *out_len);
/* out[i] = sum_j scalars[j] * poly_j[i], i < out_len (lens[j] <= out_len): the linearisation / ft polynomial of
 * kimchi/src/prover.rs:1180-1260 (f = sum of evaluated-constant x column polynomial terms, minus Z_H(zeta) t). */
int kh_poly_lincomb_dev(int field, const uint64_t *const *polys_dev, const size_t *lens, const uint64_t *scalars, size_t m,
                        uint64_t *out_dev, size_t out_len);
int kh_b_init_dev(int field, const uint64_t *elm, size_t k, const uint64_t evalscale[4], size_t padded_len, uint64_t *out_dev);
int kh_evaluate_chunks_dev(int field, const uint64_t *coeffs_dev, size_t len, size_t chunk_size, size_t num_chunks,
                           const uint64_t *points, size_t npts, uint64_t *out);
/* all polynomials of a proof in one launch: polynomial j has lens[j] coefficients and num_chunks[j] chunks; out receives, polynomial
 * after polynomial, npts x num_chunks[j] values (out_j[p][c]). */
int kh_evaluate_chunks_batch_dev(int field, const uint64_t *const *polys_dev, const size_t *lens, const size_t *num_chunks, size_t m,
                                 size_t chunk_size, const uint64_t *points, size_t npts, uint64_t *out);
int kh_divide_by_vanishing_poly_dev(int field, const uint64_t *f_dev, size_t len, unsigned log2_n, uint64_t *q_dev, uint64_t *r_dev);

/* ---- constraint expressions over resident columns (first slice) ----
 * kh_expr_evaluations_dev = Expr::evaluations (kimchi/src/circuits/expr.rs:1938-2190) for an expression lowered to the
 * reference's reverse Polish form (PolishToken, expr.rs:815-836, produced by Expr::to_polish); the machine is
 * PolishToken::evaluate (expr.rs:856-937) run per row.  tokens: ntok pairs (opcode, argument):
 *   KH_TOK_CONST k   push constants[k]     (Literal / Challenge / Mds / EndoCoefficient, resolved by the caller)
 *   KH_TOK_CELL  a   push column[a >> 1] at this row (a & 1 = 0, Curr) or the next one (a & 1 = 1, Next):
 *                    element (stride * i + next * next_shift) mod col_len  (the SubEvals rule, expr.rs:1972-1987;
 *                    d8 columns into a d8 result: stride 1, next_shift 8; into a d4 result: stride 2, next_shift 8)
 *                    -- VanishesOnZeroKnowledgeAndPreviousRows and UnnormalizedLagrangeBasis(i) are such columns
 *   KH_TOK_DUP, KH_TOK_POW n, KH_TOK_ADD, KH_TOK_MUL, KH_TOK_SUB, KH_TOK_STORE, KH_TOK_LOAD i   as in the reference
 *   (SkipIf / SkipIfNot are resolved by the caller: feature flags are fixed per index).
 * out_dev[i] (= or +=, `accumulate`) the value of the expression on row i, i < rows.  A malformed program (stack
 * underflow, Load before Store, final stack length != 1: ExprError::EmptyStack / the assert at expr.rs:935) returns
 * KH_E_INVALID before anything is launched. */
enum { KH_TOK_CONST = 0, KH_TOK_CELL = 1, KH_TOK_DUP = 2, KH_TOK_POW = 3, KH_TOK_ADD = 4, KH_TOK_MUL = 5, KH_TOK_SUB = 6,
       KH_TOK_STORE = 7, KH_TOK_LOAD = 8 };
int kh_expr_evaluations_dev(int field, const uint32_t *tokens, size_t ntok, const uint64_t *const *cols_dev, const size_t *col_len,
                            size_t ncols, const uint64_t *constants, size_t nconsts, size_t rows, unsigned stride,
                            unsigned next_shift, int accumulate, uint64_t *out_dev);

/* The gate library (kimchi/src/circuits/polynomials/{poseidon,complete_add,varbasemul,endosclmul,endomul_scalar,xor,rot}.rs, range_check/, foreign_field_add/,
 * foreign_field_mul/) as COMPILED kernels: index(gate) * sum_i alpha^i constraint_i per row (prover.rs:824-868), the same value the token program of that gate
 * gives through kh_expr_evaluations_dev, several times faster (no operand stack in LDS).  cols_dev: 31 device columns of col_len elements -- witness 0..14,
 * coefficients 15..29, the gate's selector 30 --, addressed like KH_TOK_CELL (stride, next_shift); constants: the table the caller builds for the gate
 * (literals, MDS entries, the endo coefficient, powers of alpha: kh_gate_num_constants entries, layout fixed per gate at build time by
 * tools/gen_gate_kernels.py -- proof_systems_amd/polish.py::gate_program returns it).  Gates are numbered 0 .. kh_gate_count() - 1; kh_gate_name gives the
 * reference's GateType name.  Two more ids follow the library, for the arguments every circuit has: "Generic" (generic.rs:100-131; witness 0..5, coefficients
 * 15..24, the generic selector 30; constants [1, alpha]) and "Permutation" (permutation.rs:225-288: alpha0 zkpm (z prod_i (w_i + gamma + x beta shift_i) -
 * z(xw) prod_i (w_i + gamma + sigma_i beta)); witness 0..6, sigma_i at 15 + i, z 22, x 23, zkpm 24; constants [gamma, beta, alpha0, beta shift_0..6]).
 * Columns an expression does not read may be any valid pointer. */
int kh_gate_count(void);
const char *kh_gate_name(int gate);
int kh_gate_num_constants(int gate);
/* The constants table of `gate` for one proof, built on the host (no device call): the protocol's literals and Poseidon MDS entries, alpha^i for the
 * gate's constraints (alpha: Montgomery limbs; may be NULL for "Generic" / "Permutation"), the endo coefficient (endo: VerifierIndex::endo, NULL if
 * the gate has none), and for "Generic" / "Permutation" the caller's per-proof values `params` in the order given above.  out: kh_gate_num_constants x 4 limbs. */
int kh_gate_constants(int field, int gate, const uint64_t alpha[4], const uint64_t endo[4], const uint64_t *params, size_t nparams, uint64_t *out);
int kh_gate_evaluations_dev(int field, int gate, const uint64_t *const *cols_dev, size_t col_len, const uint64_t *constants, size_t nconsts,
                            size_t rows, unsigned stride, unsigned next_shift, int accumulate, uint64_t *out_dev);

/* ---- scans, batch inversion, division by a linear factor: the permutation argument's vector steps ----
 * kh_field_scan_dev: in-place inclusive prefix (reverse = 0) or suffix (reverse = 1) scan under + or *
 *   (the running product z[j+1] = z[j] * ..., kimchi/src/circuits/polynomials/permutation.rs:556-563).
 * kh_batch_inversion_dev = ark_ff::batch_inversion (permutation.rs:533): non-zero entries inverted, zeros kept.
 * kh_divide_by_linear_dev: f = q (x - a) + rem (the two boundary quotients (z - 1)/(x - 1), (z - 1)/(x - sid[n - zk])
 *   of perm_quot, permutation.rs:300-327): q_dev gets len - 1 coefficients, rem = f(a) comes back to the host
 *   (the reference returns an error when it is non-zero). */
enum { KH_SCAN_ADD = 0, KH_SCAN_MUL = 1 };
int kh_field_scan_dev(int field, int op, int reverse, uint64_t *data_dev, size_t n);
int kh_batch_inversion_dev(int field, uint64_t *v_dev, size_t n);
int kh_divide_by_linear_dev(int field, const uint64_t *f_dev, size_t len, const uint64_t a[4], uint64_t *q_dev, uint64_t rem[4]);
/* The same division with the remainder left ON THE DEVICE (rem_dev: 4 limbs, nullable) -- nothing waits for the stream -- and a deferred check:
 * kh_check_equal_dev sets bit `bit` of *flags_dev (a uint32_t in device memory the caller zeroed) when any of the n elements at v_dev differs
 * from `expect` (4 limbs; NULL = zero).  A device-resident prover queues its invariants (remainders are zero, the accumulators end at 1:
 * prover.rs:913-917, permutation.rs:301-321, 566-568) behind the steps that produce them and reads the one word at the end of the proof, instead
 * of stalling the stream once per check; both are asynchronous on the library's main stream like the other vector steps. */
int kh_divide_by_linear_async_dev(int field, const uint64_t *f_dev, size_t len, const uint64_t a[4], uint64_t *q_dev, uint64_t *rem_dev);
int kh_check_equal_dev(const uint64_t *v_dev, size_t n, const uint64_t *expect, uint32_t *flags_dev, unsigned bit);

/* ---- challenge polynomials (verifier side) ----
 * kh_b_poly_coefficients = b_poly_coefficients (poly-commitment/src/commitment.rs:464-476) for k challenge sets of
 * `rounds` challenges each (k x rounds x 4 limbs, Montgomery): out[j][i] = prod_{bit b of i} chals[j][rounds-1-b],
 * k x 2^rounds x 4 limbs.
 * kh_batch_dlog_accumulator_generate (utils.rs:282-312): comms[j] = <b_poly_coefficients(chals_j), g>, coefficient
 * vectors built and consumed on the device (one batched MSM over the resident tables).
 * kh_batch_dlog_accumulator_check (utils.rs:212-273): *ok = [ sum_j r^j (C_j - <s_j, g>) == 0 ]; the reference draws
 * r from OsRng inside the function, here the caller passes it (Montgomery limbs).  Size mismatches that are
 * assert!s in the reference return KH_E_INVALID. */
int kh_b_poly_coefficients(int field, const uint64_t *chals, unsigned rounds, size_t k, uint64_t *out);
int kh_batch_dlog_accumulator_generate(kh_srs_t *srs, size_t num_comms, const uint64_t *chals, size_t chals_len,
                                       uint64_t *out_xy, uint8_t *out_inf);
int kh_batch_dlog_accumulator_check(kh_srs_t *srs, const uint64_t *comms_xy, const uint8_t *comms_inf, size_t k,
                                    const uint64_t *chals, size_t chals_len, const uint64_t r[4], int *ok);
/* The single MSM of the batch verifier SRS::verify (poly-commitment/src/ipa.rs:301-502):
 *   sum_i sg_weights[i] <b_poly_coefficients(chals_i), g>  +  sum_j extra_scalars[j] extra_j   == 0 ?
 * The k challenge polynomials (k x rounds challenges, 2^rounds = SRS size) are expanded on the device and multiplied
 * into the resident tables (ipa.rs:402-420); the proof-specific terms (H, sg, U, L/R, the commitments, delta with the
 * scalars of ipa.rs:405-470, all computed by the caller next to its sponge) go through the ad-hoc MSM. */
int kh_ipa_verify_msm(kh_srs_t *srs, const uint64_t *chals, size_t chals_len, const uint64_t *sg_weights, size_t k,
                      const uint64_t *extra_xy, const uint8_t *extra_inf, const uint64_t *extra_scalars, size_t m, int *is_zero);

/* ---- the folding loop of SRS::open on the device (poly-commitment/src/ipa.rs:929-1007) ----
 * The caller keeps the sponge and the RNG (ipa.rs:940-941, 966-971); everything between two squeezes runs here
 * on device-resident vectors, so a round costs one host<->device hop of 2 points + 1 challenge:
 *
 *   kh_ipa_begin(srs, a = p.coeffs (a_len <= srs size, zero-padded as ipa.rs:918-920), b = b_init
 *                (padded_length entries), u_base = U)                                  -> state
 *   per round:  kh_ipa_round_lr(state, rand_l, rand_r) -> (L, R)      ipa.rs:943-961
 *               kh_ipa_round_fold(state, u_pre)        -> (u, u^-1)   ipa.rs:972-1006, u = u_pre.to_field(endo_r)
 *   kh_ipa_finish(state) -> a0 = a[0], b0 = b[0], sg = g[0]                             ipa.rs:1009-1018
 *
 * L, R and sg are the reference's group elements (bit-identical affine coordinates); the basis is never folded:
 * round j's L / R are MSMs over the SRS's resident window tables with the scalars a (x) (challenge tensor), and
 * sg = <challenge tensor, G>.  The blinding base H is the handle's (kh_srs_set_blinding_base);
 * the SRS size must be a power of two; one opening at a time per SRS handle (KH_E_INVALID otherwise).
 * All field elements Montgomery limbs; u_pre as in kh_ipa_fold_points_endo. */
typedef struct kh_ipa kh_ipa_t;
int kh_ipa_begin(kh_srs_t *srs, const uint64_t *a, size_t a_len, const uint64_t *b, size_t b_len,
                 const uint64_t u_base_xy[8], kh_ipa_t **out);
/* same with a and b already in device memory (e.g. the outputs of kh_combine_polys_dev / kh_b_init_dev) */
int kh_ipa_begin_dev(kh_srs_t *srs, const uint64_t *a_dev, size_t a_len, const uint64_t *b_dev, size_t b_len,
                     const uint64_t u_base_xy[8], kh_ipa_t **out);
int kh_ipa_rounds_left(const kh_ipa_t *st);
int kh_ipa_round_lr(kh_ipa_t *st, const uint64_t rand_l[4], const uint64_t rand_r[4], uint64_t lr_xy[16], uint8_t lr_inf[2]);
int kh_ipa_round_fold(kh_ipa_t *st, const uint64_t chal[2], uint64_t u_out[4], uint64_t u_inv_out[4]);
int kh_ipa_finish(kh_ipa_t *st, uint64_t a0[4], uint64_t b0[4], uint64_t sg_xy[8], uint8_t *sg_inf);
void kh_ipa_free(kh_ipa_t *st);
/* Everything SRS::open does after combine_polys / b_init (ipa.rs:898-1060) in ONE call: absorbs shift_scalar(<a, b>), derives
 * U = to_group(challenge_fq()), runs the log2(n) rounds (L / R MSMs on the device, absorb, challenge, folds of a, b and of the
 * challenge tensor), then delta, c, z1, z2.  `sponge`: the prover's Fq-sponge (fq_sponge_before_evaluations, prover.rs:1193),
 * advanced exactly as the reference advances it.  `blinders`: what the reference draws from its RNG, in its order -- (rand_l,
 * rand_r) per round, then d, r_delta -- 2 log2(n) + 2 scalar-field elements.  blinding_factor: the combined blinder returned
 * by combine_polys.  Outputs: lr (log2(n) x 2 points), delta, z1, z2, sg: the OpeningProof (ipa.rs:1175-1191). */
int kh_ipa_open(kh_srs_t *srs, const uint64_t *a_dev, size_t a_len, const uint64_t *b_dev, size_t b_len, const uint64_t combined_inner_product[4],
                const uint64_t blinding_factor[4], kh_sponge_t *sponge, const uint64_t *blinders, size_t blinders_len,
                uint64_t *lr_xy, uint8_t *lr_inf, uint64_t delta_xy[8], uint8_t *delta_inf, uint64_t z1[4], uint64_t z2[4], uint64_t sg_xy[8], uint8_t *sg_inf);

/* PolyComm::multi_scalar_mul (poly-commitment/src/commitment.rs:350-394): m commitments with num_chunks[i] chunks each
 * (chunks_xy / chunks_inf: the chunk lists concatenated, sum(num_chunks) points), m scalars (Montgomery).
 * out chunk j = sum_{i : num_chunks[i] > j} scalars[i] * com_i.chunks[j]; *out_count = max(num_chunks), or 1 point
 * at infinity when m == 0 (the reference's empty case).  out_xy / out_inf need room for max(num_chunks, 1) points. */
int kh_polycomm_multi_scalar_mul(int curve, const uint64_t *chunks_xy, const uint8_t *chunks_inf, const size_t *num_chunks, size_t m,
                                 const uint64_t *scalars, uint64_t *out_xy, uint8_t *out_inf, size_t *out_count);

/* ---- commitment wrappers (host logic of the SRS trait over the MSM kernels) ----
 * kh_commit_non_hiding = SRS::commit_non_hiding (poly-commitment/src/ipa.rs:638-683):
 * coefficients (len x 4 limbs, Montgomery) are split into ceil(len / srs_size) chunks,
 * chunk j = sum_i coeffs[j*srs_size + i] * g[i]; the zero polynomial (all coefficients zero or
 * len == 0) commits to ONE point at infinity; the result is padded with infinities up to
 * num_chunks (never truncated).  out_xy / out_inf must have room for
 * max(num_chunks, ceil(len / srs_size), 1) entries; *out_count receives the number written. */
int kh_commit_non_hiding(kh_srs_t *srs, const uint64_t *coeffs, size_t len, size_t num_chunks,
                         uint64_t *out_xy, uint8_t *out_inf, size_t *out_count);
/* kh_commit_evaluations_non_hiding = SRS::commit_evaluations_non_hiding (ipa.rs:706-728 +
 * PolyComm::multi_scalar_mul, commitment.rs:350-394): evals live on a domain of size evals_len
 * (a power of two, >= 2^log2_domain; sub-sampled with stride evals_len >> log2_domain), and are
 * committed against the registered Lagrange basis; one output point per basis chunk.
 * A domain larger than the evaluations' domain is the reference's panic: returns KH_E_INVALID. */
int kh_commit_evaluations_non_hiding(kh_srs_t *srs, unsigned log2_domain, const uint64_t *evals, size_t evals_len,
                                     uint64_t *out_xy, uint8_t *out_inf, size_t *out_count);
/* Blinding base h of the SRS (ipa::SRS::h); defaults to SRS::create's h (ipa.rs:765-772). */
int kh_srs_set_blinding_base(kh_srs_t *srs, const uint64_t h_xy[8]);
int kh_srs_get_blinding_base(const kh_srs_t *srs, uint64_t h_xy[8]);
/* kh_mask_custom = SRS::mask_custom (ipa.rs:605-622): out_j = blinders_j * h + com_j on the host
 * (one scalar multiplication per chunk: "negligible, stays on host").  A length mismatch is
 * CommitmentError::BlindersDontMatch: returns KH_E_BLINDERS. */
#define KH_E_BLINDERS (-5)
int kh_mask_custom(kh_srs_t *srs, const uint64_t *com_xy, const uint8_t *com_inf, size_t com_len,
                   const uint64_t *blinders /* Montgomery */, size_t blinders_len,
                   uint64_t *out_xy, uint8_t *out_inf);
/* omega_{2^k} of Radix2EvaluationDomain::new(2^k) (kimchi/src/circuits/domains.rs:40-69), Montgomery. */
int kh_domain_generator(int field, unsigned log2_n, uint64_t out[4]);

/* ---- NTT -------------------------------------------------------------------
 * Replaces Radix2EvaluationDomain::{fft_in_place, ifft_in_place} behind
 * Evaluations::interpolate[_by_ref] (kimchi/src/prover.rs:289,377,433,567,614,665,
 * 907,1163; circuits/polynomials/permutation.rs:571; poly-commitment/src/utils.rs:195-196).
 * data: batch x 2^log2_n x 4 limbs, Montgomery, natural order in and out, in place;
 * inverse != 0 includes the 1/N scaling (ark-poly ifft). */
int kh_ntt(int field, uint64_t *data, unsigned log2_n, int inverse, size_t batch);
/* Tuning knob (process-wide): the largest sub-transform of one pass is 2^max_logr points (4..10; 0 = the default, 9; KH_NTT_MAX_LOGR at start-up).
 * A transform of 2^k points runs in ceil(k / max_logr) passes; results are identical for every setting (the tests run the parity cases under 8, 9, 10). */
int kh_ntt_set_max_logr(unsigned max_logr);

/* DensePolynomial::evaluate_over_domain_by_ref(d8/d4) (kimchi/src/circuits/
 * constraints.rs:490-495; prover.rs:436,617,668): n = 2^log2_n coefficients,
 * zero-extended to n << log2_blowup and forward-transformed.
 * coeffs: batch x n x 4 limbs; out: batch x (n << log2_blowup) x 4 limbs. */
int kh_lde(int field, const uint64_t *coeffs, unsigned log2_n, unsigned log2_blowup,
           uint64_t *out, size_t batch);

/* ---- device-resident variants (no PCIe in the loop) -----------------------
 * Same operations on buffers that already live in HBM (hipMalloc'd by the caller or by
 * kh_dev_alloc).  These are what the benchmark times ("inputs already resident in HBM")
 * and what a prover that keeps witness columns on the device would call.
 * Ordering: kh_ntt_dev / kh_lde_dev / kh_coset_ntt_dev / kh_dev_copy and the vector steps whose result stays on the device
 * (kh_expr_evaluations_dev, kh_poly_lincomb_dev, kh_combine_polys_dev, kh_b_init_dev, kh_field_scan_dev, kh_batch_inversion_dev,
 * kh_divide_by_vanishing_poly_dev) return as soon as their kernels are queued on the library's main stream; their host
 * arguments (token programs, pointer tables, constants) are copied before the call returns.  Every
 * other entry point that consumes a device buffer either runs on that same stream or waits for it on the device
 * (MSMs on another pipeline slot's stream wait for an event recorded on the main stream), so a producer followed by a
 * consumer needs no kh_sync in between.  Host-visible results (points, evaluations) are complete when the call
 * returns; kh_sync() is only needed before the HOST reads a device buffer through its own means. */
int kh_dev_alloc(void **ptr, size_t bytes);
int kh_dev_free(void *ptr);
int kh_dev_copy(void *dst_dev, const void *src_dev, size_t bytes);   /* device-to-device, asynchronous on the library's main stream */
int kh_dev_memset_zero(void *dst_dev, size_t bytes);
int kh_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int kh_dev_download(void *dst_host, const void *src_dev, size_t bytes);
/* count 32-byte records (field elements) at dst_dev set to `value`; asynchronous on the main stream, the value travels in the kernel's arguments */
int kh_dev_fill_elements(uint64_t *dst_dev, const uint64_t value[4], size_t count);
/* `rows` runs of `width` bytes, `src_pitch` / `dst_pitch` bytes apart: the witness columns of a prover ([Vec<F>; 15], each shorter than
 * the domain: kimchi/src/prover.rs:254-266) go into their padded device columns in one transfer. */
int kh_dev_upload_2d(void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t width, size_t rows);
/* The same transfer WITHOUT waiting for the work queued on the main stream first: for a destination that nothing queued reads or writes (a freshly allocated
 * column).  Returns when the bytes are on the device; kernels queued earlier keep running underneath it -- kh_prove sends the witness in column groups this way,
 * each group's interpolation and extension running under the next group's transfer. */
int kh_dev_upload_2d_unordered(void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch, size_t width, size_t rows);
int kh_msm_batch_dev(kh_srs_t *srs, int basis, unsigned chunk, size_t offset,
                     const uint64_t *scalars_dev, size_t n, size_t k, int scalars_are_montgomery,
                     uint64_t *out_xy /* host, k x 8 */, uint8_t *out_is_inf /* host, k */);
/* Pipelined form of kh_msm_batch_dev: kh_msm_submit enqueues all device work on one of the
 * library's four MSM pipeline slots and returns at once; kh_msm_wait blocks for that job and
 * finishes it (XYZZ -> affine on the host).  With jobs in flight the sort of one MSM and the
 * latency-bound tail (bucket reduction) of another run underneath the bucket accumulation of a third.
 * At most KH_MSM_SLOTS (4) un-waited tickets: a further submit by the thread that holds them all returns KH_E_INVALID at once; when
 * some of them are OTHER threads' (more provers than slots) it BLOCKS until one of those is waited for (back-pressure, no time-out);
 * only when every busy slot's owner is itself blocked in a submit -- nobody left to call kh_msm_wait -- it returns KH_E_INVALID. */
#define KH_MSM_SLOTS 4
int kh_msm_submit(kh_srs_t *srs, int basis, unsigned chunk, size_t offset,
                  const uint64_t *scalars_dev, size_t n, size_t k, int scalars_are_montgomery,
                  uint64_t *ticket);
int kh_msm_wait(uint64_t ticket, uint64_t *out_xy /* host, k x 8 */, uint8_t *out_is_inf /* host, k */);
/* The same pipeline from HOST scalars -- what SRS::commit_non_hiding(&DensePolynomial) hands over (poly-commitment/src/ipa.rs:638-683): the upload is cut
 * into chunks on the calling thread's copy stream and each chunk's digit pass is queued behind it, so with two MSMs in flight the PCIe transfer of MSM i + 1
 * runs underneath the accumulation of MSM i.  Returns when the scalars have been taken (pageable or pinned memory: the caller may reuse the buffer at once);
 * the job itself keeps running; kh_msm_wait collects it. */
int kh_msm_submit_host(kh_srs_t *srs, int basis, unsigned chunk, size_t offset,
                       const uint64_t *scalars /* host, k x n x 4 limbs */, size_t n, size_t k, int scalars_are_montgomery,
                       uint64_t *ticket);
int kh_ntt_dev(int field, uint64_t *data_dev, unsigned log2_n, int inverse, size_t batch);
int kh_lde_dev(int field, const uint64_t *coeffs_dev, unsigned log2_n, unsigned log2_blowup,
               uint64_t *out_dev, size_t batch);
/* One coset of an extension: out[r][i] = f_r(shift * w_n^i), i < n = 2^log2_n, for `batch` coefficient vectors (out may
 * alias coeffs).  The d8 extension of kh_lde_dev is 8 such cosets interleaved, lde[8 i + r] = coset_r[i] with
 * shift = w_{8n}^r: with one coset per GPU every later row-wise step of the quotient (expressions, z(x w) = the next row
 * of the same coset) stays local to a GPU.  Asynchronous on the main stream like kh_ntt_dev. */
int kh_coset_ntt_dev(int field, const uint64_t *coeffs_dev, unsigned log2_n, const uint64_t shift[4], uint64_t *out_dev, size_t batch);
int kh_sync(void);

/* ---- instrumentation -------------------------------------------------------
 * Device time (ms, HIP events on the library's own stream) of the last kh_msm*_dev /
 * kh_ntt*_dev / kh_lde*_dev call on this thread, split per phase.  names/ms arrays of
 * capacity cap; returns the number of phases written. */
int kh_last_timings(const char **names, float *ms, int cap);
/* Process-wide event counters (how often a path ran): "spread_retry" (an opening round's MSM met a hot bucket and was re-run with the hot-bucket
 * kernels), "fused_retry" (the one-launch sort gave up), "graph_replay" / "graph_capture" (replayed / captured launch sequences), "rebase_launch" /
 * "rebase_switch" / "rebase_abandon" (openings that started materialising the folded basis of their late rounds, switched over to it, gave it up),
 * "rebased_rounds" (rounds that ran over a materialised basis).  Unknown names read 0. */
uint64_t kh_counter(const char *name);

/* Test hooks (field ops on the device; op: 0 mul, 1 add, 2 sub, 3 to_mont, 4 from_mont,
 * 5 sqr, 6 neg).  Used by the parity tests to pin the device arithmetic itself. */
int kh_debug_field_op(int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* Test hook of the opening's rebase (csrc/rebase.hip; the folded basis of poly-commitment/src/ipa.rs:985-1003 after several rounds at once):
 * out[i] = sum_{q < Q} coef[q] * g[q N + i] for i < N = n / Q (N a multiple of 64), from the handle's c = 16 window tables; coef Montgomery, out affine.
 * *out_fail != 0: an output was the point at infinity (the rebase is then abandoned; out is incomplete). */
int kh_debug_rebase_points(kh_srs_t *srs, const uint64_t *coef, size_t Q, uint64_t *out_xy, uint32_t *out_fail);
/* Test hook of the GLV split the folded basis's MSMs use (csrc/msm.hip glv_split; constants: tools/gen_glv_params.py): for n CANONICAL scalars of the given
 * scalar field, out[10 i .. 10 i + 9] = |k1| (4 x 32-bit limbs), |k2| (4), sign of k1, sign of k2 with k = k1 + k2 lambda (mod r), lambda = the scalar-field
 * endomorphism eigenvalue of kh_endos (endo_r).  kh_scalar_challenge_to_field's curve ids pick the field: KH_FIELD_FP = Vesta's scalars, KH_FIELD_FQ = Pallas's. */
int kh_debug_glv_split(int scalar_field, const uint64_t *scalars, size_t n, uint32_t *out);
/* Point ops on the device through the XYZZ formulas: op 0 = P + Q (affine in, affine out),
 * op 1 = 2P, op 2 = P + Q via the mixed addition. inf flags in/out. */
int kh_debug_point_op(int curve, int op, const uint64_t *p_xy, const uint8_t *p_inf,
                      const uint64_t *q_xy, const uint8_t *q_inf,
                      uint64_t *out_xy, uint8_t *out_inf, size_t n);

/* ---- one process per GPU: the path's only collective, over RCCL / xGMI ----
 * A point-range-sharded MSM (BASELINE config 4) exchanges ONE partial point per rank: RCCL has no elliptic-curve reduction, so the "all-reduce" is an
 * all-gather of the partials and a local fold on every rank.  kh_comm_unique_id on one rank, its 128 bytes handed to the others by any means (file,
 * socket, the launcher's store), then kh_comm_init on every rank (collective; binds the communicator to the calling thread's current device).
 * librccl.so is loaded with dlopen at that point -- single-GPU users never need it (KH_E_NOTFOUND if it is absent).
 * kh_comm_allgather_points: out = rank 0's k points, rank 1's k points, ... (host buffers).  kh_msm_allreduce: this rank's shard (its kh_srs_t over its
 * point range, kh_srs_create_device_range) times its slice of the scalars, all-gathered and folded: every rank gets the whole MSM. */
typedef struct kh_comm kh_comm_t;
#define KH_COMM_ID_BYTES 128
int kh_comm_unique_id(uint8_t id[128]);
int kh_comm_init(int world_size, int rank, const uint8_t id[128], kh_comm_t **out);
void kh_comm_free(kh_comm_t *comm);
int kh_comm_world_size(const kh_comm_t *comm);
int kh_comm_rank(const kh_comm_t *comm);
int kh_comm_allgather_points(kh_comm_t *comm, const uint64_t *xy, const uint8_t *inf, size_t k, uint64_t *out_xy, uint8_t *out_inf);
int kh_msm_allreduce(kh_comm_t *comm, kh_srs_t *shard, const uint64_t *scalars, size_t n, int scalars_are_montgomery, uint64_t out_xy[8],
                     uint8_t *out_is_inf);

/* ---- the lookup argument's `sorted` step (kimchi/src/circuits/lookup/constraints.rs:90-194), host code as in the reference ----
 * table: the first lookup_rows = n - zk_rows - 1 entries of the combined table (4 limbs each, host memory); values: max_per_row columns of looked-up
 * joint values, column s at values + 4 s value_stride, lookup_rows entries each (a row with fewer lookups than max_per_row holds the dummy value 0 in
 * the remaining slots).  out: max_per_row + 1 columns of lookup_rows + 1 values: every table entry repeated (times looked up + 1), in table order,
 * snake layout (consecutive columns share one element, odd columns reversed).  A value that is not in the table: KH_E_INVALID and *bad_row = its row. */
int kh_lookup_sorted(const uint64_t *table, size_t lookup_rows, const uint64_t *values, size_t value_stride, size_t max_per_row, uint64_t *out,
                     size_t *bad_row);

/* ---- ProverProof::create as ONE native call (kimchi/src/prover.rs:187-1515, the part this library accelerates end to end) ----
 * The host loop of the prover -- witness columns -> commitments -> z -> quotient -> evaluations -> opening, with the transcript -- written
 * against the entry points above, so that a Rust / C caller pays neither an interpreter nor 60 FFI crossings per proof.  Scope: everything
 * create_recursive takes: generic + the five library gates + the optional gates (RangeCheck0/1, Rot64, Xor16
 * as a plain gate, ForeignFieldAdd/Mul), public inputs, any num_chunks, previous challenges (kh_prove_recursive), lookups into fixed and runtime tables (kh_prover_index_attach_lookup, kh_prover_index_attach_runtime_tables, kh_prove_full).  (proof_systems_amd/prover.py runs the same protocol from Python and
 * covers lookups / runtime tables / recursion; tests/test_gpu_native_prover.py: both give the same proof, field element for field element.)
 *
 * kh_prover_index_new: the caller has built the index columns on the device (ProverIndex of prover_index.rs:30-70; column order below) on the
 * device of `srs`, whose Lagrange basis for 2^log2_n is registered (kh_srs_compute_lagrange).  Three blocks of columns of n = 2^log2_n
 * (d1_dev: evaluations on the domain; dc_dev: coefficient forms) resp. 8n (d8_dev: evaluations on the 8x extended domain) elements:
 *     column 0..14 coefficients | 15 generic selector | 16 sid (omega^j) | 17..23 sigma_0..6 | 24..28 selectors of Poseidon, CompleteAdd,
 *     VarBaseMul, EndoMul, EndoMulScalar | 29.. the n_optional optional-gate selectors (optional_gates: their kh_gate ids, in column order);
 *     dc_dev / d8_dev hold two more columns after those: x and the permutation vanishing polynomial (permutation.rs:107-118).
 *   live_mask: bit k set = the circuit HAS rows of the k-th selector column after the generic one (24 + k): only those gates' constraints are
 *   evaluated unless KH_PROVE_ALL_GATES asks for the reference's behaviour (prover.rs:824-868 evaluates every always-present gate type).
 *   shifts: the 7 permutation shifts (Shifts::new); digest: VerifierIndex::digest (an element of the curve's base field).
 *   The index keeps the pointers (no copy): they must outlive it.
 * kh_prove: witness = 15 columns x rows x 4 Montgomery limbs on the host (rows + zk_rows <= n; the zero-knowledge rows are drawn here), or
 *   witness_dev = the padded 15 x n columns already on the device (then no zero-knowledge rows are drawn: the caller has).
 *   randomness: the field elements the reference draws from its RNG, IN ITS ORDER (kh_prove_randomness_count of them; uniform scalar-field
 *   elements, Montgomery limbs) -- zero-knowledge rows per column from the last row backwards (prover.rs:254-266, host witness only), 15 x
 *   num_chunks witness blinders, z's two random rows, num_chunks blinders of z, 7 num_chunks of t, then the opening's (rand_l, rand_r) per round,
 *   d, r_delta (ipa.rs:940-941, 1036) -- or NULL: drawn from the operating system's generator (getrandom).
 * kh_proof_section: a view into the proof (valid until kh_proof_free).  Point sections give limbs = count x 8 (affine x | y) and flags =
 *   count infinity flags; element sections give limbs = count x 4 and flags = NULL. */
typedef struct kh_prover_index kh_prover_index_t;
typedef struct kh_proof kh_proof_t;
#define KH_PROVE_CHECK 1          /* assert the intermediate invariants (z ends at 1, zero remainders) */
#define KH_PROVE_ALL_GATES 2      /* evaluate the constraints of every always-present gate type, as the reference does */
#define KH_PROVE_EAGER_CHECK 8    /* with KH_PROVE_CHECK: read every invariant back where it is produced and fail THERE, as the reference's ProverError returns do
                                   * (prover.rs:913-917, permutation.rs:566-568, lookup/constraints.rs:325-331) -- one stream stall per check; by default the checks
                                   * ride on the device and are read once after the opening (same errors, reported at the end of the call) */
#define KH_PROVE_SHARED_CONTEXT 4 /* stay on the device's shared library context (default: a private one for the call, kh_private_context_begin) */
#define KH_PROOF_W_COMM 0         /* 15 x num_chunks points */
#define KH_PROOF_Z_COMM 1         /* num_chunks points */
#define KH_PROOF_T_COMM 2         /* 7 num_chunks points */
#define KH_PROOF_PUBLIC_COMM 3    /* num_chunks points (verifier side; not part of the serialised proof) */
#define KH_PROOF_EVALS 4          /* per polynomial num_chunks values at zeta, then num_chunks at zeta omega; polynomials: z, generic selector, the five
                                     selectors, w x 15, coefficients x 15, sigma x 6, the optional selectors in column order */
#define KH_PROOF_PUBLIC_EVALS 5   /* num_chunks at zeta, num_chunks at zeta omega */
#define KH_PROOF_FT_EVAL1 6
#define KH_PROOF_LR 7             /* (L, R) per round: 2 log2(srs size) points */
#define KH_PROOF_DELTA 8
#define KH_PROOF_Z1_Z2 9
#define KH_PROOF_SG 10
#define KH_PROOF_CHALLENGES 11    /* beta, gamma, alpha, zeta, v (polyscale), u (evalscale); with lookups a seventh: the joint combiner */
#define KH_PROOF_LOOKUP_SORTED_COMM 12   /* (max lookups per row + 1) x num_chunks points; empty without lookups */
#define KH_PROOF_LOOKUP_AGGREG_COMM 13   /* num_chunks points */
#define KH_PROOF_LOOKUP_RUNTIME_COMM 14  /* num_chunks points; empty without runtime tables */
int kh_prover_index_new(kh_srs_t *srs, unsigned log2_n, unsigned zk_rows, unsigned public_inputs, const uint64_t *d1_dev, const uint64_t *dc_dev,
                        const uint64_t *d8_dev, const int *optional_gates, size_t n_optional, unsigned live_mask, const uint64_t *shifts,
                        const uint64_t digest[4], kh_prover_index_t **out);
/* Adds the lookup constraint system of the index (LookupConstraintSystem, lookup/index.rs:189-311; fixed tables, no runtime tables): patterns = the
 * ids of the lookup patterns the circuit uses -- 0 Xor, 1 Lookup, 2 RangeCheck, 3 ForeignFieldMul --, increasing; per pattern its selector column as
 * d1 evaluations, coefficient form and d8 evaluations; the concatenated table columns and (nullable) the table-id column as d1 evaluations; and the d8
 * evaluations of the three row-set atoms the constraints use (expr.rs:883-893): VanishesOnZeroKnowledgeAndPreviousRows, UnnormalizedLagrangeBasis(0),
 * UnnormalizedLagrangeBasis(-zk_rows - 1).  The digest given to kh_prover_index_new must cover the lookup index.  kh_prove then runs the lookup argument
 * (joint combiner, combined table, sorted columns via kh_lookup_sorted, aggregation, the lookup constraints on d8, the extra evaluations and openings);
 * KH_PROOF_EVALS carries, after the polynomials listed above: sorted x (max_per_row + 1), aggregation, combined table, one selector per pattern;
 * the randomness grows by (max_per_row + 1) (zk_rows + num_chunks) after the witness blinders and zk_rows + num_chunks before z's two rows. */
int kh_prover_index_attach_lookup(kh_prover_index_t *index, const int *patterns, size_t n_patterns, const uint64_t *const *selectors_d1,
                                  const uint64_t *const *selectors_c, const uint64_t *const *selectors_d8, const uint64_t *const *table_cols_d1,
                                  size_t n_table_cols, const uint64_t *table_ids_d1, const uint64_t *const *atoms_d8);
/* Runtime tables (lookup/runtime_tables.rs, index.rs:241-311): `length` rows of the combined table, starting at row `offset`, whose second column arrives
 * with each proof (kh_prove_full: runtime_values, all runtime tables' data concatenated in the index's order); selector = the runtime-table selector column
 * (1 outside the runtime rows, 0 on them and on the zero-knowledge rows) as d1 evaluations, coefficient form, d8 evaluations.  After
 * kh_prover_index_attach_lookup.  KH_PROOF_EVALS then carries runtime table + runtime selector between the combined table and the pattern selectors; the
 * randomness grows by zk_rows + num_chunks right after the witness blinders. */
int kh_prover_index_attach_runtime_tables(kh_prover_index_t *index, const uint64_t *selector_d1, const uint64_t *selector_c, const uint64_t *selector_d8,
                                          size_t offset, size_t length);
void kh_prover_index_free(kh_prover_index_t *index);
size_t kh_prove_randomness_count(const kh_prover_index_t *index, int witness_on_host);
int kh_prove(kh_prover_index_t *index, const uint64_t *witness, size_t rows, const uint64_t *witness_dev, const uint64_t *randomness,
             size_t n_random, unsigned flags, kh_proof_t **out);
/* create_recursive with previous challenges (RecursionChallenge, proof.rs:117-131; prover.rs:276-279, 1212-1262): n_prev accumulators, the j-th with
 * prev_rounds[j] challenges (prev_chals: all of them concatenated, Montgomery limbs) and a commitment of prev_comm_chunks[j] chunks (prev_comm_xy /
 * prev_comm_inf: all chunks concatenated).  2^rounds must be the SRS size (one chunk) or twice it (two chunks). */
int kh_prove_recursive(kh_prover_index_t *index, const uint64_t *witness, size_t rows, const uint64_t *witness_dev, const uint64_t *randomness,
                       size_t n_random, unsigned flags, const uint64_t *prev_chals, const unsigned *prev_rounds, const uint64_t *prev_comm_xy,
                       const uint8_t *prev_comm_inf, const size_t *prev_comm_chunks, size_t n_prev, kh_proof_t **out);
/* everything create_recursive takes: previous challenges and (n_runtime = the index's runtime rows, else 0) the runtime tables' second column */
int kh_prove_full(kh_prover_index_t *index, const uint64_t *witness, size_t rows, const uint64_t *witness_dev, const uint64_t *randomness, size_t n_random,
                  unsigned flags, const uint64_t *prev_chals, const unsigned *prev_rounds, const uint64_t *prev_comm_xy, const uint8_t *prev_comm_inf,
                  const size_t *prev_comm_chunks, size_t n_prev, const uint64_t *runtime_values, size_t n_runtime, kh_proof_t **out);
int kh_proof_section(const kh_proof_t *proof, int section, const uint64_t **limbs, const uint8_t **flags, size_t *count);
int kh_proof_phase_seconds(const kh_proof_t *proof, double *seconds, size_t cap);   /* witness_upload, witness_commit, z, quotient, evaluations, opening */
void kh_proof_free(kh_proof_t *proof);

/* ---- Fiat-Shamir sponges (host side) --------------------------------------
 * The transcript of kimchi's prover / verifier: Kimchi Poseidon (width 3, rate 2, 55 full rounds, x^7) under
 * DefaultFqSponge (poseidon/src/sponge.rs:228-412) and DefaultFrSponge (kimchi/src/plonk_sponge.rs:36-57).  Host code, no
 * GPU needed.  kind KH_SPONGE_FQ = the sponge over `curve`'s BASE field that absorbs commitments and squeezes challenges of
 * the scalar field; KH_SPONGE_FR = the sponge over `curve`'s SCALAR field that absorbs evaluations.  Field elements are
 * 4 Montgomery limbs as everywhere; a challenge is the raw 128-bit value (2 limbs), to be mapped with
 * kh_scalar_challenge_to_field where the reference wraps it in a ScalarChallenge. */
#define KH_SPONGE_FQ 0
#define KH_SPONGE_FR 1
int kh_sponge_new(int kind, int curve, kh_sponge_t **out);
int kh_sponge_clone(const kh_sponge_t *s, kh_sponge_t **out);           /* fq_sponge.clone() (prover.rs:1193) */
void kh_sponge_free(kh_sponge_t *s);
int kh_sponge_absorb_g(kh_sponge_t *s, const uint64_t *xy, const uint8_t *inf /*nullable*/, size_t n);   /* FqSponge::absorb_g */
int kh_sponge_absorb(kh_sponge_t *s, const uint64_t *x, size_t n);      /* elements of the sponge's own field: absorb_fq / FrSponge::absorb(_multiple) */
int kh_sponge_absorb_fr(kh_sponge_t *s, const uint64_t *x, size_t n);   /* FqSponge::absorb_fr: scalar-field elements into the base-field sponge */
int kh_sponge_challenge(kh_sponge_t *s, uint64_t chal[2]);              /* 128-bit challenge, raw */
int kh_sponge_challenge_field(kh_sponge_t *s, uint64_t out[4]);         /* the same as a scalar-field element (beta, gamma) */
int kh_sponge_squeeze_field(kh_sponge_t *s, uint64_t out[4]);           /* challenge_fq / digest_fq / FrSponge::digest */
int kh_sponge_digest(kh_sponge_t *s, uint64_t out[4]);                  /* FqSponge::digest: as a scalar-field element, 0 if it does not fit */

#ifdef __cplusplus
}
#endif
#endif /* KIMCHI_HIP_H */
