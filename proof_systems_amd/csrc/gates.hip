// gates.hip -- the gate library's combined constraints as compiled kernels (kh_gate_evaluations_dev).
//
// prover.rs:824-868 evaluates, for every gate type, index(gate) * sum_i alpha^i constraint_i over the d8 columns.  csrc/expr.hip can run any
// such expression as a token program, but its operand stack and Store / Load slots live in LDS, which caps the Poseidon / VarBaseMul / EndoMulScalar
// programs at one wave per SIMD (16-30 G products/s).  The gate library is protocol data fixed at build time, so tools/gen_gate_kernels.py lowers
// the SAME expression DAGs (proof_systems_amd/polish.py) to straight-line functions (gates_gen.inc): every intermediate value a register-resident
// field element, common sub-expressions once, no interpreter.  One kernel per (gate, field); thread per row; columns 0..14 = witness, 15..29 =
// coefficients, 30 = the gate's selector; the constants table is the one polish.gate_program returns (literals, MDS, endo, powers of alpha).
// Bound: VALU issue (products of 254 instructions), as everywhere in this library.
#include "common.hpp"
#include "field.cuh"
#include "msm.hpp"
#include "host_ec.hpp"

namespace kh {

struct GateArgs {
    const u64* cols[31];
    const u64* consts;
    size_t rows, len;
    u32 stride, next_shift;
    int accumulate;
    u64* out;
};
template <class F>
struct GateCtx {
    const GateArgs& a;
    size_t i0, i1;                                       // element index of this row / of the next row in every column
    __device__ __forceinline__ Fe<F> cell(int c, int nxt) const { return Fe<F>::load(a.cols[c] + 4 * (nxt ? i1 : i0)); }
    __device__ __forceinline__ Fe<F> cst(int k) const { return Fe<F>::load(a.consts + 4 * k); }
};

#include "gates_gen.inc"

#define KH_GATE_KERNEL(ID, NAME)                                                                              \
    template <class F>                                                                                        \
    __global__ void __launch_bounds__(128) k_gate_##NAME(GateArgs a) {                                        \
        const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;                                              \
        const size_t row = i < a.rows ? i : a.rows - 1;                                                       \
        size_t i0 = (size_t)a.stride * row, i1 = i0 + a.next_shift;                                           \
        if (i1 >= a.len) i1 -= a.len;                                                                         \
        const GateCtx<F> g{a, i0, i1};                                                                        \
        Fe<F> v = gate_##NAME<F>(g);                                                                          \
        if (i < a.rows) {                                                                                     \
            if (a.accumulate) v = add<F>(Fe<F>::load(a.out + 4 * i), v);                                      \
            v.store(a.out + 4 * i);                                                                           \
        }                                                                                                     \
    }
KH_FOR_EACH_GATE(KH_GATE_KERNEL)

int gate_count() { return GATE_COUNT; }
const char* gate_name(int gate) { return gate >= 0 && gate < GATE_COUNT ? GATE_NAMES[gate] : nullptr; }
int gate_num_constants(int gate) { return gate >= 0 && gate < GATE_COUNT ? GATE_NCONST[gate] : -1; }

// the constants table of one gate for one proof, on the host (gates_gen.inc: GATE_CONST_TABLE): literals of the protocol, powers of alpha, the endo
// coefficient, the caller's per-proof values -- so that a caller needs no copy of the expression builder to drive kh_gate_evaluations_dev
int gate_constants(int field, int gate, const uint64_t* alpha, const uint64_t* endo, const uint64_t* params, size_t nparams, uint64_t* out) {
    KH_REQUIRE(gate >= 0 && gate < GATE_COUNT, "unknown gate id %d", gate);
    KH_REQUIRE(field == KH_FIELD_FP || field == KH_FIELD_FQ, "unknown field %d", field);
    const khost::Fld F(field == KH_FIELD_FP ? 0 : 1);
    const GateConst* rc = GATE_CONST_TABLE[gate];
    khost::fe apow[64]; int have = 0;                     // alpha^1 .. alpha^have
    for (int k = 0; k < GATE_NCONST[gate]; k++) {
        khost::fe v;
        switch (rc[k].kind) {
            case 0: { khost::fe c; for (int i = 0; i < 4; i++) c.l[i] = rc[k].lit[field == KH_FIELD_FP ? 0 : 1][i]; v = F.to_mont(c); break; }
            case 1: {
                KH_REQUIRE(alpha, "gate %s needs alpha", GATE_NAMES[gate]);
                KH_REQUIRE(rc[k].arg >= 1 && rc[k].arg < 64, "bad power in the constants recipe");
                if (!have) { memcpy(&apow[1], alpha, 32); have = 1; }
                while (have < rc[k].arg) { apow[have + 1] = F.mul(apow[have], apow[1]); have++; }
                v = apow[rc[k].arg]; break;
            }
            case 2: KH_REQUIRE(endo, "gate %s needs the endo coefficient", GATE_NAMES[gate]); memcpy(&v, endo, 32); break;
            default:
                KH_REQUIRE(params && (size_t)rc[k].arg < nparams, "gate %s takes %d per-proof values, got %zu", GATE_NAMES[gate], rc[k].arg + 1, nparams);
                memcpy(&v, params + 4 * (size_t)rc[k].arg, 32); break;
        }
        memcpy(out + 4 * (size_t)k, &v, 32);
    }
    return KH_OK;
}

#define g_gate_consts (kh::ctx().scratch("gate_consts"))

int gate_run(Context& C, int field, int gate, const uint64_t* const* cols_dev, size_t len, const uint64_t* consts, size_t nconsts, size_t rows,
             unsigned stride, unsigned next_shift, int accumulate, uint64_t* out_dev) {
    KH_REQUIRE(gate >= 0 && gate < GATE_COUNT, "unknown gate id %d", gate);
    KH_REQUIRE((int)nconsts == GATE_NCONST[gate], "gate %s takes %d constants (polish.gate_program), got %zu", GATE_NAMES[gate], GATE_NCONST[gate], nconsts);
    KH_REQUIRE(len > 0 && (size_t)stride * (rows ? rows - 1 : 0) < len && next_shift < len, "rows * stride must not exceed the column length (%zu rows, stride %u, length %zu)", rows, stride, len);
    if (rows == 0) return KH_OK;
    int rc;
    if ((rc = g_gate_consts.reserve(GATE_COUNT * 64 * 32))) return rc;
    // a slot of the constants scratch per gate: consecutive gate launches of one proof do not overwrite each other's table while queued
    u64* d_consts = g_gate_consts.as<u64>() + (size_t)gate * 64 * 4;
    if ((rc = C.stage_upload(d_consts, {{consts, nconsts * 32}}))) return rc;
    GateArgs a{};
    for (int c = 0; c < 31; c++) { KH_REQUIRE(cols_dev[c], "column %d is null", c); a.cols[c] = cols_dev[c]; }
    a.consts = d_consts; a.rows = rows; a.len = len; a.stride = stride; a.next_shift = next_shift; a.accumulate = accumulate; a.out = out_dev;
    hipStream_t s = C.stream;
    dim3 grid((unsigned)((rows + 127) / 128));
    C.timer.begin(s);
    switch (gate) {
#define KH_GATE_LAUNCH(ID, NAME)                                                                                           \
        case ID:                                                                                                               \
            if (field == KH_FIELD_FP) hipLaunchKernelGGL((k_gate_##NAME<FpParams>), grid, dim3(128), 0, s, a);                 \
            else hipLaunchKernelGGL((k_gate_##NAME<FqParams>), grid, dim3(128), 0, s, a);                                      \
            break;
        KH_FOR_EACH_GATE(KH_GATE_LAUNCH)
        default: break;
    }
    KH_HIP(hipGetLastError());
    C.timer.mark("gate", s);
    return KH_OK;
}

}  // namespace kh
