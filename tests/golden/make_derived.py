#!/usr/bin/env python3
"""Derived golden vectors (NOT from the reference, which holds no NTT known-answer test -- SURVEY 8c):
outputs of the big-int oracle (oracle/pasta.py: naive DFT by definition, textbook group law) on small
seeded inputs, committed so that the C oracle and the HIP path are also checked against fixed bytes.
    python tests/golden/make_derived.py"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pasta as P  # noqa: E402

rnd = random.Random(20260925)
out = {"_generated_by": "tests/golden/make_derived.py (oracle/pasta.py, definition-level arithmetic)", "ntt": [], "lde": [], "msm": []}
for name, F in (("Fp", P.Fp), ("Fq", P.Fq)):
    for k in (1, 3, 4):
        a = [rnd.randrange(F.p) for _ in range(1 << k)]
        out["ntt"].append({"field": name, "log2_n": k, "input": [hex(v) for v in a],
                           "forward": [hex(v) for v in P.dft_naive(F, a, k)],
                           "inverse": [hex(v) for v in P.dft_naive(F, a, k, inverse=True)]})
    c = [rnd.randrange(F.p) for _ in range(4)]
    out["lde"].append({"field": name, "log2_n": 2, "log2_blowup": 3, "coeffs": [hex(v) for v in c],
                       "evals": [hex(v) for v in P.dft_naive(F, c + [0] * 28, 5)]})
for name, c in (("vesta", P.VESTA), ("pallas", P.PALLAS)):
    pts = [c.srs_g(i) for i in range(12)]
    sc = [rnd.randrange(c.scalar.p) for _ in range(9)] + [0, 1, c.scalar.p - 1]
    r = c.msm_naive(pts, sc)
    out["msm"].append({"curve": name, "bases": "SRS::create g_0..g_11", "scalars": [hex(v) for v in sc],
                       "result": [hex(r[0]), hex(r[1])]})
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "derived_vectors.json"), "w") as f:
    json.dump(out, f, indent=1)
print("ok")
