// expr.hip -- row-parallel evaluation of a compiled constraint expression over device-resident columns: the GPU
// counterpart of Expr::evaluations (kimchi/src/circuits/expr.rs:1938-2190) for an expression already lowered to the
// reference's reverse Polish form (PolishToken, expr.rs:815-836; Expr::to_polish).  SURVEY 8f rank 2, first slice:
// the interpreter.  The token machine is PolishToken::evaluate (expr.rs:856-937) run once per ROW instead of once
// per evaluation point: Cell(col, Curr|Next) reads column[(stride * i + next * shift) mod len] exactly like the
// SubEvals arm of evaluations (expr.rs:1972-1987), Constant / Challenge / Mds / EndoCoefficient / Literal are entries
// of a constants table the caller resolves, VanishesOnZeroKnowledgeAndPreviousRows and UnnormalizedLagrangeBasis are
// columns the prover precomputes anyway (constraints.rs precomputations, prover.rs l0_1), SkipIf / SkipIfNot are
// resolved by the caller (feature flags are per index).
//
// One thread per row; the token stream is wave-uniform (no divergence).  The operand stack lives in LDS, laid out
// [slot][word][thread] (conflict-free), with the top of the stack held in registers: a binary operation costs one LDS
// read, a push one LDS write.  Cost model = the expression's multiplications (254 VALU instructions each): VALU-issue
// bound like every kernel here once an expression has more than a handful of products per column read.
#include "common.hpp"
#include "field.cuh"
#include "msm.hpp"

namespace kh {

static constexpr int EXPR_T = 128;          // threads per block

template <class F>
__device__ __forceinline__ void lds_put(u32* lds, int slot, const Fe<F>& v) {
#pragma unroll
    for (int k = 0; k < 8; k++) lds[(slot * 8 + k) * EXPR_T + threadIdx.x] = v.v[k];
}
template <class F>
__device__ __forceinline__ Fe<F> lds_get(const u32* lds, int slot) {
    Fe<F> r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = lds[(slot * 8 + k) * EXPR_T + threadIdx.x];
    return r;
}

template <class F>
__global__ void __launch_bounds__(EXPR_T)
k_expr(const u32* __restrict__ prog, u32 ntok, const u64* const* __restrict__ cols, const u64* __restrict__ col_len,
       const u64* __restrict__ consts, size_t rows, u32 stride, u32 next_shift, int stack_slots, int accumulate, int nowrap, u64* __restrict__ out) {
    extern __shared__ u32 lds[];
    const size_t i = (size_t)blockIdx.x * EXPR_T + threadIdx.x;
    const size_t row = i < rows ? i : rows - 1;          // idle lanes shadow the last row (uniform control flow)
    Fe<F> top = Fe<F>::zero();
    int sp = 0;                                           // elements below the register-held top
    int ncache = 0;
    bool have_top = false;
    for (u32 t = 0; t < ntok; t++) {
        const u32 op = prog[2 * t], arg = prog[2 * t + 1];
        switch (op) {
            case KH_TOK_CONST:
            case KH_TOK_CELL:
            case KH_TOK_LOAD:
            case KH_TOK_DUP: {
                if (have_top) { lds_put<F>(lds, sp, top); sp++; }
                if (op == KH_TOK_CONST) top = Fe<F>::load(consts + 4 * arg);
                else if (op == KH_TOK_CELL) {
                    const u32 c = arg >> 1;
                    const size_t len = col_len[c];
                    size_t idx = (size_t)stride * row + ((arg & 1u) ? next_shift : 0u);
                    // a 64-bit remainder is ~100 VALU instructions per cell read; when stride * rows <= len and the shift < len (every call
                    // of the provers: `nowrap`, decided on the host) the index wraps at most once
                    if (nowrap) { if (idx >= len) idx -= len; } else idx %= len;
                    top = Fe<F>::load(cols[c] + 4 * idx);
                } else if (op == KH_TOK_LOAD) top = lds_get<F>(lds, stack_slots + (int)arg);
                /* DUP: top unchanged */
                have_top = true;
                break;
            }
            case KH_TOK_POW: {
                Fe<F> base = top, acc = Fe<F>::one();
                for (u32 e = arg; e; e >>= 1) { if (e & 1u) acc = mul<F>(acc, base); if (e > 1) base = sqr<F>(base); }
                top = acc;
                break;
            }
            case KH_TOK_ADD: { sp--; top = add<F>(lds_get<F>(lds, sp), top); break; }
            case KH_TOK_MUL: { sp--; top = mul<F>(lds_get<F>(lds, sp), top); break; }
            case KH_TOK_SUB: { sp--; top = sub<F>(lds_get<F>(lds, sp), top); break; }
            case KH_TOK_STORE: { lds_put<F>(lds, stack_slots + ncache, top); ncache++; break; }
            default: break;
        }
    }
    if (i < rows) {
        if (accumulate) top = add<F>(Fe<F>::load(out + 4 * i), top);
        top.store(out + 4 * i);
    }
}

#define g_expr_tab (kh::ctx().scratch("expr_tab"))

// validates the program the way PolishToken::evaluate would fail (EmptyStack, final stack length != 1, Load of a
// value never stored) and returns the LDS slots it needs
static int expr_check(const uint32_t* prog, size_t ntok, size_t ncols, size_t nconsts, int* stack_slots, int* cache_slots) {
    long depth = 0, maxd = 0, ncache = 0;
    for (size_t t = 0; t < ntok; t++) {
        const uint32_t op = prog[2 * t], arg = prog[2 * t + 1];
        switch (op) {
            case KH_TOK_CONST: KH_REQUIRE(arg < nconsts, "token %zu: constant %u of %zu", t, arg, nconsts); depth++; break;
            case KH_TOK_CELL: KH_REQUIRE((arg >> 1) < ncols, "token %zu: column %u of %zu", t, arg >> 1, ncols); depth++; break;
            case KH_TOK_DUP: KH_REQUIRE(depth >= 1, "token %zu: Dup on an empty stack", t); depth++; break;
            case KH_TOK_POW: KH_REQUIRE(depth >= 1, "token %zu: Pow on an empty stack", t); break;
            case KH_TOK_ADD: case KH_TOK_MUL: case KH_TOK_SUB:
                KH_REQUIRE(depth >= 2, "token %zu: binary operation on a stack of %ld (ExprError::EmptyStack)", t, depth); depth--; break;
            case KH_TOK_STORE: KH_REQUIRE(depth >= 1, "token %zu: Store on an empty stack", t); ncache++; break;
            case KH_TOK_LOAD: KH_REQUIRE((long)arg < ncache, "token %zu: Load(%u) before its Store", t, arg); depth++; break;
            default: KH_REQUIRE(false, "token %zu: unknown opcode %u", t, op);
        }
        if (depth > maxd) maxd = depth;
    }
    KH_REQUIRE(depth == 1, "the program leaves %ld values on the stack (assert_eq!(stack.len(), 1), expr.rs:935)", depth);
    *stack_slots = (int)maxd; *cache_slots = (int)ncache;
    return KH_OK;
}

int expr_run(Context& C, int field, const uint32_t* prog, size_t ntok, const uint64_t* const* cols_dev, const size_t* col_len, size_t ncols,
             const uint64_t* consts, size_t nconsts, size_t rows, unsigned stride, unsigned next_shift, int accumulate, uint64_t* out_dev) {
    int slots = 0, cslots = 0;
    int rc = expr_check(prog, ntok, ncols, nconsts, &slots, &cslots); if (rc) return rc;
    const size_t lds = (size_t)(slots + cslots) * 32 * EXPR_T;
    KH_REQUIRE(lds <= 160 * 1024, "expression needs %d stack + %d cache slots: more than the 160 KB of LDS holds for %d rows", slots, cslots, EXPR_T);
    int nowrap = 1;
    for (size_t c = 0; c < ncols; c++) {
        KH_REQUIRE(col_len[c] > 0 && cols_dev[c], "column %zu is empty", c);
        if (!((size_t)stride * (rows ? rows - 1 : 0) < col_len[c] && next_shift < col_len[c])) nowrap = 0;
    }
    if (rows == 0) return KH_OK;
    const size_t bytes = ntok * 8 + ncols * 16 + nconsts * 32 + 64;
    if ((rc = g_expr_tab.reserve(bytes))) return rc;
    char* base = g_expr_tab.as<char>();
    char* d_cols = base; char* d_len = d_cols + ncols * 8; char* d_consts = d_len + ncols * 8; char* d_prog = d_consts + nconsts * 32;
    std::vector<u64> len64(col_len, col_len + ncols);
    hipStream_t s = C.stream;
    if ((rc = C.stage_upload(base, {{cols_dev, ncols * 8}, {len64.data(), ncols * 8}, {consts, nconsts * 32}, {prog, ntok * 8}}))) return rc;
    if (C.once("expr_attr")) {
        KH_HIP(hipFuncSetAttribute((const void*)k_expr<FpParams>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KH_HIP(hipFuncSetAttribute((const void*)k_expr<FqParams>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    dim3 grid((unsigned)((rows + EXPR_T - 1) / EXPR_T));
    C.timer.begin(s);
    if (field == KH_FIELD_FP)
        hipLaunchKernelGGL((k_expr<FpParams>), grid, dim3(EXPR_T), lds, s, (const u32*)d_prog, (u32)ntok, (const u64* const*)d_cols, (const u64*)d_len,
                           (const u64*)d_consts, rows, (u32)stride, (u32)next_shift, slots, accumulate, nowrap, out_dev);
    else
        hipLaunchKernelGGL((k_expr<FqParams>), grid, dim3(EXPR_T), lds, s, (const u32*)d_prog, (u32)ntok, (const u64* const*)d_cols, (const u64*)d_len,
                           (const u64*)d_consts, rows, (u32)stride, (u32)next_shift, slots, accumulate, nowrap, out_dev);
    KH_HIP(hipGetLastError());
    C.timer.mark("expr", s);
    return KH_OK;                 // asynchronous on the main stream, like kh_ntt_dev
}

}  // namespace kh
