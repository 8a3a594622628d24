#!/usr/bin/env python3
"""Soak: many proofs through kh_prove, first from one thread with a seeded generator (every proof must equal the first), then from four threads with
the library's randomness (every 25th proof goes to the oracle verifier).  Catches rare races in the latency-path kernels (grid barriers, graph replay,
slot hand-over) that single proofs do not.  Usage: tools/soak.py [proofs_single] [proofs_per_thread]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
from proof_systems_amd import prover
from oracle import kimchi as K, pasta as P, views as V
khip.init(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
logn = 14
ixs = [prover.bench_circuit_index(khip.VESTA, logn) for _ in range(4)]
F = ixs[0].F
wit = np.tile(F.limbs(1), (15, (1 << logn) - 10, 1))
ref = V.device_views(ixs[0], prover.create_proof_native(ixs[0], wit, np.random.default_rng(5)))[2]
t0 = time.perf_counter()
for i in range(n1):
    pr = V.device_views(ixs[0], prover.create_proof_native(ixs[0], wit, np.random.default_rng(5), check=False))[2]
    assert pr == ref, "proof %d differs from the first" % i
print(f"{n1} seeded proofs identical ({1e3 * (time.perf_counter() - t0) / n1:.2f} ms each incl. the comparison)")
errs, kept = [], []
def work(t):
    try:
        prover.create_proof_native(ixs[t], wit, None)
        for i in range(n2):
            p = prover.create_proof_native(ixs[t], wit, None, check=(i % 10 == 0))
            if i % 25 == 0:
                kept.append((t, p))
    except BaseException as e:
        errs.append(e)
th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
t0 = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t0
assert not errs, errs
for t, p in kept:
    c, vix, pr = V.device_views(ixs[t], p)
    assert K.verify(c, vix, pr, None, vix["h"], P.StdRng(bytes([9] * 32)), final_msm=V.final_msm_c(c, ixs[t].srs.get_g(), ixs[t].size)), "a proof was rejected"
print(f"4 threads x {n2} proofs: {4 * n2 / dt:.1f} proofs/s, {len(kept)} of them verified")
