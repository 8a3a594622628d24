#!/usr/bin/env python3
"""Device time of every gate's constraint program (proof_systems_amd/polish.py) on 2^19 rows of d8 columns: what `seconds_all_gates` adds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
from proof_systems_amd import prover, polish as OP
khip.init(0)
F = prover.Fld(khip.FP)
n8 = 1 << 19
rng = np.random.default_rng(1)
cols = []
for _ in range(31):
    s = rng.integers(0, 1 << 64, size=(n8, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 61) - 1)
    cols.append(khip.DevBuf(n8 * 32).upload(s))
out = khip.DevBuf(n8 * 32)
endo = F.value(khip.endos(1)[0])
for name in OP.GATES:
    toks, consts = OP.gate_program(name, F.p, 12345, selector_col=30, mds=OP.POSEIDON_MDS[0], endo=endo)
    nmul = sum(1 for o, a in toks if o == OP.TOK_MUL) + sum(bin(a).count("1") + a.bit_length() - 2 for o, a in toks if o == OP.TOK_POW)
    ncell = sum(1 for o, a in toks if o == OP.TOK_CELL)
    ts = []
    for _ in range(5):
        khip.expr_evaluations_dev(khip.FP, toks, cols, [n8] * 31, F.limbs_many(consts), n8, out, stride=1, next_shift=8)
        khip.sync(); ts.append(sum(ms for k, ms in khip.last_timings() if k == "expr"))
    tc = []
    gid = khip.gate_ids().get(name)
    out2 = khip.DevBuf(n8 * 32)
    for _ in range(5):
        khip.gate_evaluations_dev(khip.FP, gid, cols, n8, F.limbs_many(consts), n8, out2, stride=1, next_shift=8)
        khip.sync(); tc.append(sum(ms for k, ms in khip.last_timings() if k == "gate"))
    same = np.array_equal(out.download((n8, 4)), out2.download((n8, 4)))
    out2.free()
    print(f"{name:16s} tokens {len(toks):5d}  products {nmul:4d}  cell reads {ncell:4d}  token machine {min(ts):7.3f} ms ({nmul * n8 / min(ts) / 1e6:5.1f} G products/s)   compiled {min(tc):7.3f} ms  equal={same}")
