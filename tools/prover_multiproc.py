#!/usr/bin/env python3
"""Upper bound for several provers on one GPU: P independent PROCESSES (own library context, streams and Python interpreter each),
every one producing proofs of the benchmark circuit back to back.  Compare with tools/prover_concurrent.py (threads in one process).
Usage: tools/prover_multiproc.py [log2_n] [processes] [proofs per process]"""
import os, subprocess, sys, time
logn = sys.argv[1] if len(sys.argv) > 1 else "16"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if len(sys.argv) > 4 and sys.argv[4] == "worker":
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import proof_systems_amd.khip as khip
    from proof_systems_amd import prover
    khip.init(0)
    ix = prover.bench_circuit_index(khip.VESTA, int(logn))
    F = prover.Fld(ix.fid)
    wit = np.tile(F.limbs(1), (15, (1 << int(logn)) - 10, 1))
    rng = np.random.default_rng(os.getpid())
    nx = prover.native_index(ix)                                  # kh_prove with the library's randomness: the same call the threads of prover_concurrent.py --native make
    for _ in range(3):
        nx.prove(witness=wit, randomness=None, flags=0)
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    for _ in range(K):
        nx.prove(witness=wit, randomness=None, flags=0)
    print(f"done {time.perf_counter() - t0:.4f}", flush=True)
    sys.exit(0)
procs = [subprocess.Popen([sys.executable, __file__, logn, str(P), str(K), "worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(P)]
for p in procs:
    assert p.stdout.readline().strip() == "ready"
t0 = time.perf_counter()
for p in procs:
    p.stdin.write("go\n"); p.stdin.flush()
times = [float(p.stdout.readline().split()[1]) for p in procs]
dt = time.perf_counter() - t0
print(f"{P} processes x {K} proofs: {P * K / dt:.1f} proofs/s = {P * K * (1 << int(logn)) / dt / 1e6:.2f} M constraints/s (per process {min(times):.3f}-{max(times):.3f} s)")
