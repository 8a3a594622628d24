#!/usr/bin/env python3
"""Proof throughput with several provers in flight on one GPU: T host threads, each with its own index / SRS handle (an opening holds its
handle's workspace), each producing proofs of the benchmark circuit back to back.  One proof is mostly latency chains (opening rounds,
transcript on the host), so independent proofs overlap.  Usage: tools/prover_concurrent.py [log2_n] [threads ...]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
from proof_systems_amd import prover
khip.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NATIVE = "--native" in sys.argv                        # kh_prove (C++ host loop, the library draws the randomness: no Python between the steps, no GIL)
counts = [int(x) for x in sys.argv[2:] if not x.startswith("--")] or [1, 2, 4]
tmax = max(counts)
ixs = [prover.bench_circuit_index(khip.VESTA, logn) for _ in range(tmax)]
F = prover.Fld(ixs[0].fid)
wit = np.tile(F.limbs(1), (15, (1 << logn) - 10, 1))
for ix in ixs:
    prover.create_proof(ix, wit, np.random.default_rng(1), check=False)
PROOFS = 6
nxs = [prover.native_index(ix) for ix in ixs] if NATIVE else []
for T in counts:
    bar = threading.Barrier(T + 1)

    def work(t):
        rng = np.random.default_rng(100 + t)
        try:
            if NATIVE:                                  # this thread's own library context: warm it (workspaces, graphs) outside the timed region
                nxs[t].prove(witness=wit, randomness=None, flags=0)
            bar.wait()
            for _ in range(PROOFS):
                if NATIVE:
                    nxs[t].prove(witness=wit, randomness=None, flags=0)
                else:
                    prover.create_proof(ixs[t], wit, rng, check=False)
            bar.wait()
        except BaseException:
            bar.abort()                  # a failing prover must not leave the others (and the GPU box) waiting at the barrier
            raise
    th = [threading.Thread(target=work, args=(t,), daemon=True) for t in range(T)]
    for t in th:
        t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in th:
        t.join()
    print(f"{T} prover(s) in flight: {T * PROOFS / dt:.1f} proofs/s = {T * PROOFS * (1 << logn) / dt / 1e6:.2f} M constraints/s  ({1e3 * dt / PROOFS:.1f} ms per proof per thread)")
