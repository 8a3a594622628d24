"""Throughput of the opening loop with T provers in flight (one host thread and one SRS handle each), 2^16."""
import sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proof_systems_amd import khip
khip.init(0)
n = 1 << 16
rng = np.random.default_rng(1)
def rs(k):
    a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
U = khip.srs_generate(0, 1 << 21, 1)[0]
chals = [int.from_bytes(rng.bytes(16), "little") for _ in range(16)]
def one(srs, a, b, r):
    op = khip.IpaOpening(srs, a, b, U)
    for ch in chals:
        op.round_lr(r[0], r[1]); op.round_fold(ch)
    op.finish(); op.free()
for T in (1, 2, 3, 4):
    ctx = [(khip.Srs.create(0, n), rs(n), rs(n), rs(2)) for _ in range(T)]
    for c in ctx: one(*c); one(*c)
    reps = 6
    def work(c):
        for _ in range(reps): one(*c)
    th = [threading.Thread(target=work, args=(c,)) for c in ctx]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{T} provers: {1e3 * dt / reps:.2f} ms per round of {T} openings = {1e3 * dt / (reps * T):.2f} ms per opening")
    for c in ctx: c[0].close()
