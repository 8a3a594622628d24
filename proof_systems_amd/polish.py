"""Reverse-Polish token streams (include/kimchi_hip.h KH_TOK_*) of the constraints this repository restates from the
reference, for callers of kh_expr_evaluations_dev that have no Rust `Expr::to_polish()` at hand (bench.py, tools/):
the double generic gate (kimchi/src/circuits/polynomials/generic.rs:83-120, argument.rs:201-214) and the `perm` part of
perm_quot (polynomials/permutation.rs:237-283)."""
TOK_CONST, TOK_CELL, TOK_DUP, TOK_POW, TOK_ADD, TOK_MUL, TOK_SUB, TOK_STORE, TOK_LOAD = range(9)


def cell(col: int, nxt: int = 0):
    return (TOK_CELL, 2 * col + nxt)


def generic_gate_tokens(w0: int, c0: int, sel: int, alpha0: int, alpha1: int):
    """index(Generic) * (alpha^a0 * constraint1 + alpha^a1 * constraint2); witness columns w0..w0+5, coefficient columns
    c0..c0+9, selector column sel, alpha powers at constants alpha0 / alpha1."""
    t = [cell(sel)]
    for g, alpha in ((0, alpha0), (1, alpha1)):
        w, c = w0 + 3 * g, c0 + 5 * g
        t += [(TOK_CONST, alpha)]
        t += [cell(c), cell(w), (TOK_MUL, 0)]
        t += [cell(c + 1), cell(w + 1), (TOK_MUL, 0), (TOK_ADD, 0)]
        t += [cell(c + 2), cell(w + 2), (TOK_MUL, 0), (TOK_ADD, 0)]
        t += [cell(c + 3), cell(w), (TOK_MUL, 0), cell(w + 1), (TOK_MUL, 0), (TOK_ADD, 0)]
        t += [cell(c + 4), (TOK_ADD, 0)]
        t += [(TOK_MUL, 0)]
        if g == 1:
            t += [(TOK_ADD, 0)]
    t += [(TOK_MUL, 0)]
    return t


def perm_quot_tokens(w0: int, s0: int, z: int, x: int, zkpm: int, gamma: int, beta: int, bshift0: int, alpha0: int, permuts: int = 7):
    """alpha0 * zkpm(x) * (z(x) prod_i (w_i + gamma + x beta shift_i) - z(x w) prod_i (w_i + gamma + sigma_i beta))."""
    t = []
    for i in range(permuts):
        t += [cell(w0 + i), (TOK_CONST, gamma), (TOK_ADD, 0), cell(x), (TOK_CONST, bshift0 + i), (TOK_MUL, 0), (TOK_ADD, 0)]
        if i:
            t += [(TOK_MUL, 0)]
    t += [cell(z), (TOK_MUL, 0)]
    for i in range(permuts):
        t += [cell(w0 + i), (TOK_CONST, gamma), cell(s0 + i), (TOK_CONST, beta), (TOK_MUL, 0), (TOK_ADD, 0), (TOK_ADD, 0)]
        if i:
            t += [(TOK_MUL, 0)]
    t += [cell(z, 1), (TOK_MUL, 0), (TOK_SUB, 0), (TOK_CONST, alpha0), (TOK_MUL, 0), cell(zkpm), (TOK_MUL, 0)]
    return t


def perm_aggreg_tokens(permuts: int = 7, w0: int = 0, s0: int = 7, sid: int = 14, gamma: int = 0, beta: int = 1, bshift0: int = 2):
    """(numerator, denominator) rows of perm_aggreg (permutation.rs:510-551): prod_i (w_i + sid beta shift_i + gamma) and
    prod_i (w_i + sigma_i beta + gamma)."""
    num, den = [], []
    for i in range(permuts):
        num += [cell(w0 + i), cell(sid), (TOK_CONST, bshift0 + i), (TOK_MUL, 0), (TOK_ADD, 0), (TOK_CONST, gamma), (TOK_ADD, 0)] + ([(TOK_MUL, 0)] if i else [])
        den += [cell(w0 + i), cell(s0 + i), (TOK_CONST, beta), (TOK_MUL, 0), (TOK_ADD, 0), (TOK_CONST, gamma), (TOK_ADD, 0)] + ([(TOK_MUL, 0)] if i else [])
    return num, den
