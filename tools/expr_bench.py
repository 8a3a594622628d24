"""kh_expr_evaluations_dev on 2^19 rows (d8 of a 2^16-row circuit): the double generic gate (17 columns, 14 products per row)
and a product-heavy synthetic expression (15 columns, sbox-like x^7 terms, ~165 products per row)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from proof_systems_amd import khip
import proof_systems_amd.polish as P
khip.init(0)
rows = 1 << 19
rng = np.random.default_rng(3)
def rs(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
cols = [khip.DevBuf(rows * 32).upload(rs(rows)) for _ in range(17)]
out = khip.DevBuf(rows * 32)
consts = rs(12)
progs = {"generic_gate": P.generic_gate_tokens(0, 6, 16, 0, 1)}
t = []
for r in range(5):                                         # 5 rounds x 3 state words: (sum mds * x^7) + rc - next
    for i in range(3):
        first = True
        for j in range(3):
            t += [(P.TOK_CELL, 2 * (3 * r + j)), (P.TOK_POW, 7), (P.TOK_CONST, 3 * i + j), (P.TOK_MUL, 0)]
            if not first: t += [(P.TOK_ADD, 0)]
            first = False
        t += [(P.TOK_CELL, 2 * ((3 * r + 3 + i) % 15) + (1 if r == 4 else 0)), (P.TOK_SUB, 0), (P.TOK_CONST, 9 + (i % 3)), (P.TOK_MUL, 0)]
        if r or i: t += [(P.TOK_ADD, 0)]
progs["poseidon_like"] = t
for name, toks in progs.items():
    nmul = sum(1 for op, a in toks if op == P.TOK_MUL) + sum(4 for op, a in toks if op == P.TOK_POW)
    ncell = sum(1 for op, a in toks if op == P.TOK_CELL)
    khip.expr_evaluations_dev(0, toks, cols, [rows] * 17, consts, rows, out)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); khip.expr_evaluations_dev(0, toks, cols, [rows] * 17, consts, rows, out); ts.append(time.perf_counter() - t0)
    khip.sync()
    dev = [ms for nm, ms in khip.last_timings() if nm == "expr"]
    ms = 1e3 * float(np.median(ts))
    print(f"{name}: {len(toks)} tokens, {nmul} products + {ncell} column reads per row; {ms:.3f} ms wall, kernel {dev} ms; "
          f"{rows * nmul / (ms * 1e-3) / 1e9:.1f} G products/s, {rows * (ncell + 1) * 32 / (ms * 1e-3) / 1e9:.0f} GB/s of column traffic")
