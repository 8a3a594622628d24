#!/usr/bin/env python3
"""Generates proof_systems_amd/csrc/gates_gen.inc: the gate library's combined constraints as STRAIGHT-LINE device code.

The token machine of csrc/expr.hip runs any caller-supplied expression, but its operand stack and Store / Load slots live in LDS (4 KB per slot
per block): the Poseidon program needs ~20 slots, which leaves one wave per SIMD and 27 G products/s of the 139 G/s the product allows
(tools/gate_expr_time.py).  The gate library is fixed protocol data, so its expressions are compiled ahead of time instead: this script builds
the same expression DAGs proof_systems_amd/polish.py lowers to tokens (same builder functions, same constants table), and emits one
`gate_<Name><F>(ctx)` function per gate in SSA form -- every node a register-resident Fe<F>, shared sub-expressions computed once, x^7 as
sqr / sqr / mul / mul.  csrc/gates.hip wraps them in kernels (`kh_gate_evaluations_dev`).  The constants table a kernel reads is exactly
`polish.gate_program(name, ...)[1]` (literals, MDS entries, the endo coefficient, powers of alpha), so the caller computes it as before.

Like the asm generators, the output is committed; tests/test_gates.py checks that it is current and tests/test_gpu_gates.py that the kernels
equal the token machine and the oracle's row machines.  Usage: python tools/gen_gate_kernels.py [--check]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proof_systems_amd import polish as OP  # noqa: E402

OUT = os.path.join(ROOT, "proof_systems_amd", "csrc", "gates_gen.inc")
P_FP = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
P_FQ = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
ALPHA = 0x1d2c3b4a59687796a5b4c3d2e1f00112233445566778899aabbccddeeff00123      # placeholder: only the LAYOUT of the constants table matters
ENDO = 0x2d33357cb532458ed3552a23a8554e5005270d29d19fc7d27b7fd22f0201b547       # (any value different from every literal)
RECIPES = {}
GATE_IDS = list(OP.GATES) + list(OP.COMPILED_EXTRA)       # the library, then the two arguments every circuit has (generic, permutation)


def build(name, p, fid):
    env = OP.Env(p, w0=0, c0=15, mds=OP.POSEIDON_MDS[fid], endo=ENDO)
    if name in OP.COMPILED_EXTRA:
        expr = OP.COMPILED_EXTRA[name](env)              # per-proof values only: env.param slots, in the order the docstring gives
    else:
        fn, count = OP.GATES[name]
        cs = fn(env)
        assert len(cs) == count
        expr = OP.combined_constraints(env, 30, cs, ALPHA)
    toks = OP.compile_tokens(env, expr)                  # enters the literals into env.consts in the token program's order
    return env, expr, toks


def emit(name):
    env, expr, toks = build(name, P_FP, 0)
    env_q, _, toks_q = build(name, P_FQ, 1)
    assert toks == toks_q and len(env.consts) == len(env_q.consts), "the constants layout must not depend on the field"
    lines, memo, counter = [], {}, [0]

    def var():
        counter[0] += 1
        return "t%d" % counter[0]

    def go(n):
        if isinstance(n, OP._Lit):
            n = env.const(n.arg)
        key = id(n)
        if n.op == OP.TOK_CONST:
            key = ("c", n.arg)
        elif n.op == OP.TOK_CELL:
            key = ("w", n.arg)
        if key in memo:
            return memo[key]
        if n.op == OP.TOK_CONST:
            v = var(); lines.append("const Fe<F> %s = g.cst(%d);" % (v, n.arg))
        elif n.op == OP.TOK_CELL:
            v = var(); lines.append("const Fe<F> %s = g.cell(%d, %d);" % (v, n.arg >> 1, n.arg & 1))
        elif n.op == OP.TOK_POW:
            x = go(n.a)
            e = n.arg
            assert e >= 1
            acc, base = None, x
            while e:                                     # square-and-multiply without the token machine's multiplication by one
                if e & 1:
                    if acc is None:
                        acc = base
                    else:
                        v = var(); lines.append("const Fe<F> %s = mul<F>(%s, %s);" % (v, acc, base)); acc = v
                e >>= 1
                if e:
                    v = var(); lines.append("const Fe<F> %s = sqr<F>(%s);" % (v, base)); base = v
            v = acc
        elif n.op == "dbl":
            x = go(n.a); v = var(); lines.append("const Fe<F> %s = add<F>(%s, %s);" % (v, x, x))
        elif n.op == "sqr":
            x = go(n.a); v = var(); lines.append("const Fe<F> %s = sqr<F>(%s);" % (v, x))
        else:
            a = go(n.a); b = go(n.b)
            f = {OP.TOK_ADD: "add", OP.TOK_MUL: "mul", OP.TOK_SUB: "sub"}[n.op]
            v = var(); lines.append("const Fe<F> %s = %s<F>(%s, %s);" % (v, f, a, b))
        memo[key] = v
        return v
    sys.setrecursionlimit(100000)
    root = go(expr)
    nmul = sum(1 for l in lines if "mul<F>" in l or "sqr<F>" in l)
    body = "\n".join("    " + l for l in lines)
    recipe = []
    powers = {f: {pow(ALPHA, i, P) : i for i in range(1, 64)} for f, P in ((0, P_FP), (1, P_FQ))}
    params = getattr(env, "param_slots", set())
    for k in range(len(env.consts)):
        a, b = env.consts[k], env_q.consts[k]
        if k in params:
            recipe.append((3, sorted(params).index(k), 0, 0))
        elif a == ENDO % P_FP and b == ENDO % P_FQ:
            recipe.append((2, 0, 0, 0))
        elif a in powers[0]:
            assert powers[1].get(b) == powers[0][a]
            recipe.append((1, powers[0][a], 0, 0))
        else:
            recipe.append((0, 0, a, b))
    RECIPES[name] = recipe
    return ("// %s: %d constraints, %d products, %d constants\ntemplate <class F>\n__device__ __forceinline__ Fe<F> gate_%s(const GateCtx<F>& g) {\n%s\n    return %s;\n}\n"
            % (name, OP.GATES[name][1] if name in OP.GATES else {"Generic": 2, "Permutation": 1}[name], nmul, len(env.consts), name, body, root)), len(env.consts)


def render():
    out = ["// GENERATED by tools/gen_gate_kernels.py from proof_systems_amd/polish.py -- do not edit.",
           "// One straight-line function per gate of the library: index(gate) * sum_i alpha^i constraint_i on one row (column numbering: witness 0..14,",
           "// coefficients 15..29, the gate's selector 30; constants table = polish.gate_program(name, ...)[1]).", ""]
    counts = []
    for name in GATE_IDS:
        src, nc = emit(name)
        out.append(src); counts.append(nc)
    out.append("#define KH_FOR_EACH_GATE(X) " + " ".join("X(%d, %s)" % (k, n) for k, n in enumerate(GATE_IDS)))
    out.append("static constexpr int GATE_COUNT = %d;" % len(GATE_IDS))
    out.append("static const char* const GATE_NAMES[GATE_COUNT] = {%s};" % ", ".join('"%s"' % n for n in GATE_IDS))
    out.append("static constexpr int GATE_NCONST[GATE_COUNT] = {%s};" % ", ".join(map(str, counts)))
    # how the caller's constants table is made (kh_gate_constants): kind 0 = a literal of the protocol (canonical value per field: small integers, 2^k,
    # the Poseidon MDS), 1 = alpha^arg, 2 = the endo coefficient, 3 = the caller's arg-th per-proof value (challenges)
    limbs = lambda v: ", ".join("0x%016xULL" % ((v >> (64 * i)) & 0xffffffffffffffff) for i in range(4))
    out.append("struct GateConst { int kind, arg; unsigned long long lit[2][4]; };")
    for name in GATE_IDS:
        rows = ["    {%d, %d, {{%s}, {%s}}}," % (k, a, limbs(x), limbs(y)) for k, a, x, y in RECIPES[name]]
        out.append("static const GateConst GATE_CONSTS_%s[] = {\n%s\n};" % (name, "\n".join(rows)))
    out.append("static const GateConst* const GATE_CONST_TABLE[GATE_COUNT] = {%s};" % ", ".join("GATE_CONSTS_" + n for n in GATE_IDS))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")
