#!/usr/bin/env python3
"""Times create_proof on a circuit of Lookup gates into two user tables (the shape of kimchi/src/tests/lookup.rs) at 2^log2_n rows: how much of a
lookup proof is host work (sorted columns, conversions) in the Python loop.  Usage: tools/lookup_prover_time.py [log2_n]"""
import os, sys, time, random
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
from proof_systems_amd import prover, lookup as LK
khip.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = 1 << logn
rnd = random.Random(21)
F = prover.Fld(khip.FP)
tsz = min(n // 4, 4096)
tables = [{"id": 0, "data": [list(range(tsz)), [0] + [rnd.randrange(F.p) for _ in range(tsz - 1)]]},
          {"id": 3, "data": [list(range(tsz // 2)), [rnd.randrange(F.p) for _ in range(tsz // 2)]]}]
ngen = 30; nlook = n - 3 - ngen - 8
co = np.zeros((ngen, 15, 4), dtype=np.uint64)
co[:, 0, :] = F.limbs(1); co[:, 4, :] = F.limbs(F.p - 7)
gates = ["Generic"] * ngen + ["Lookup"] * nlook + ["Zero"] * (n - 3 - ngen - nlook)
rows = ngen + nlook
wit = [[0] * rows for _ in range(15)]
for r in range(ngen):
    wit[0][r] = 7
for r in range(ngen, rows):
    t = tables[rnd.randrange(2)]
    wit[0][r] = t["id"]
    for i in range(3):
        e = rnd.randrange(len(t["data"][0]))
        wit[2 * i + 1][r], wit[2 * i + 2][r] = t["data"][0][e], t["data"][1][e]
srs = khip.Srs.create(khip.VESTA, n)
ix = prover.ProverIndex(khip.VESTA, logn, co, srs=srs)
ix.attach_lookup(LK.LookupIndex(khip.FP, gates, tables, logn))
w = np.stack([F.limbs_many(c) for c in wit])
prover.create_proof(ix, w, np.random.default_rng(8))
best = None
for _ in range(3):
    t = {}
    prover.create_proof(ix, w, np.random.default_rng(8), timings=t, check=False)
    if best is None or t["total"] < best["total"]:
        best = t
print(f"2^{logn} rows, {nlook} Lookup gates: " + "  ".join(f"{k} {1e3 * v:.2f} ms" for k, v in best.items()))

prover.create_proof_native(ix, w, np.random.default_rng(8))
best = None
for _ in range(3):
    t = {}
    prover.create_proof_native(ix, w, np.random.default_rng(8), timings=t, check=False)
    if best is None or t["total"] < best["total"]:
        best = t
print(f"kh_prove: " + "  ".join(f"{k} {1e3 * v:.2f} ms" for k, v in best.items()))
