#!/usr/bin/env python3
"""Host <-> device transfer rate of kh_dev_upload / kh_dev_download on PAGEABLE memory (what the drop-in boundary hands over).
Measured on the MI355X box: ~55 GB/s both ways, also for buffers that were never transferred before -- hipMemcpy's own staging
is at the PCIe 5 x16 rate here; a multi-threaded staging layer written for this library in round 2 was slower (34-46 GB/s) and was dropped.
Usage: tools/copy_bench.py [MB]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
khip.init(0)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 31
a = np.random.default_rng(0).integers(0, 1 << 62, size=(mb << 20) // 8, dtype=np.uint64)
b = khip.DevBuf(a.nbytes)
for name, fn in (("upload", lambda: b.upload(a)), ("download", lambda: b.download(a.shape))):
    fn(); best = 1e9
    for _ in range(7):
        t = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t)
    print(f"{name}: {mb} MB in {1e3 * best:.2f} ms = {a.nbytes / best / 1e9:.1f} GB/s")
assert (b.download(a.shape) == a).all()
# fresh buffers: what a caller that builds its vectors per proof hands over (every page touched once, never transferred before)
best = 1e9
for _ in range(5):
    c = a.copy()
    t = time.perf_counter(); b.upload(c); best = min(best, time.perf_counter() - t)
print(f"upload of a fresh array: {1e3 * best:.2f} ms = {a.nbytes / best / 1e9:.1f} GB/s")
best = 1e9
for _ in range(5):
    t = time.perf_counter(); r = b.download(a.shape); best = min(best, time.perf_counter() - t)      # download() allocates its result
print(f"download into a fresh array: {1e3 * best:.2f} ms = {a.nbytes / best / 1e9:.1f} GB/s")
