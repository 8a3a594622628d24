#!/usr/bin/env python3
"""Throughput of T independent provers (one host thread, one SRS handle, one library context each) on one GPU through kh_prove -- bench.py's
`prover.concurrent` with more proofs per thread.  Usage: tools/concurrent_provers.py [threads=4] [proofs_per_thread=20]
Environment switches worth alternating: KH_IPA_GRAPH=1 (opening rounds replay a captured graph), KH_NO_DONE_FLAG=1 (completion by event only)."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip  # noqa: E402
from proof_systems_amd import prover  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20
log_n = 16
khip.init(0)
khip.set_phase_timers(False)
ixs = [prover.bench_circuit_index(khip.VESTA, log_n) for _ in range(T)]
wit = np.tile(ixs[0].F.limbs(1), (15, (1 << log_n) - 10, 1))
for j in ixs:
    prover.create_proof(j, wit, np.random.default_rng(2), check=False)
nxs = [prover.native_index(j) for j in ixs]
bar = threading.Barrier(T + 1)


def run(t):
    nxs[t].prove(witness=wit, randomness=None, flags=0)
    bar.wait()
    for _ in range(per):
        nxs[t].prove(witness=wit, randomness=None, flags=0)
    bar.wait()


def cpu_stat():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0)), int(d.get("nr_throttled", 0))
    except OSError:
        return 0, 0, 0


for rep in range(3):
    th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for t_ in th:
        t_.start()
    bar.wait(); c0 = cpu_stat(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0; c1 = cpu_stat()
    for t_ in th:
        t_.join()
    print(f"{T} provers in flight, {per} proofs each: {T * per / dt:.1f} proofs/s   (host: {(c1[1] - c0[1]) / 1e6 / dt:.1f} cores busy on average, "
          f"cgroup throttled {c1[2] - c0[2]} times / {(c1[0] - c0[0]) / 1e3:.1f} ms in {dt * 1e3:.0f} ms)", flush=True)
