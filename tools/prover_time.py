#!/usr/bin/env python3
"""Times proof_systems_amd.prover.create_proof on the benchmark circuit (kimchi/src/bench.rs) and prints the phase split."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
from proof_systems_amd import prover
khip.init(0)
if not os.environ.get("KH_IPA_TIMING"):
    khip.set_phase_timers(False)          # the Python loop runs on the shared context: as a latency-critical caller would
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t0 = time.perf_counter()
ix = prover.bench_circuit_index(khip.VESTA, logn)
print(f"index (SRS::create + Lagrange basis + column forms + commitments): {time.perf_counter() - t0:.3f} s")
F = prover.Fld(ix.fid)
wit = np.tile(F.limbs(1), (15, (1 << logn) - 10, 1))
rng = np.random.default_rng(1)
prover.create_proof(ix, wit, rng)
for check in (True, False):
    best = None
    for _ in range(5):
        t = {}
        khip.sync()
        prover.create_proof(ix, wit, rng, timings=t, check=check)
        if best is None or t["total"] < best["total"]:
            best = t
    print(f"check={check}: " + "  ".join(f"{k} {1e3 * v:.2f} ms" for k, v in best.items()), f" -> {(1 << logn) / best['total'] / 1e6:.2f} M constraints/s")
if "--native" in sys.argv or os.environ.get("KH_TIME_NATIVE"):      # the same proof through kh_prove (host loop in C++, csrc/prover.cpp)
    prover.create_proof_native(ix, wit, rng)
    for check in (True, False):
        best = None
        for _ in range(5):
            t = {}
            khip.sync()
            t0 = time.perf_counter()
            prover.create_proof_native(ix, wit, rng, timings=t, check=check)
            t["wall_incl_python_glue"] = time.perf_counter() - t0
            if best is None or t["total"] < best["total"]:
                best = t
        print(f"native check={check}: " + "  ".join(f"{k} {1e3 * v:.2f} ms" for k, v in best.items()), f" -> {(1 << logn) / best['total'] / 1e6:.2f} M constraints/s")
    sw = khip.counter("rebase_switch")
    if sw:
        print(f"opening rounds over the folded basis: {khip.counter('rebased_rounds') / sw:.2f} per opening that switched ({sw} of {khip.counter('rebase_launch')} launched; {khip.counter('rebase_abandon')} abandoned)")
