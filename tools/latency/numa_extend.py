#!/usr/bin/env python3
"""Is the host-buffer path's slow mode (16 concurrent eightfold extensions: 5.5 ms in some processes, 11 in others) a NUMA placement effect?  Runs the same
16 threaded kh_lde calls with the process (and therefore the first touch of its pageable buffers) bound to the CPUs of each NUMA node in turn.
Usage: numa_extend.py [node] [--register] [--reps N]
--register (round 6): the callers' input and output buffers are registered with the runtime first (hipHostRegister: page-locked, the copies are
plain DMA instead of the runtime's staged pageable path) -- does the repetition-to-repetition slow mode (5-6 ms against 9-15) belong to the pageable path?"""
import glob
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
allowed = sorted(os.sched_getaffinity(0))
print("NUMA nodes:", {k: "%d cpus (%d..%d)" % (len(v), v[0], v[-1]) for k, v in nodes.items()}, " allowed to this process:", len(allowed))
for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print("GPU", f.split("/")[4], "numa_node", open(f).read().strip())
register = "--register" in sys.argv
args = [a for a in sys.argv[1:] if a != "--register"]
reps_arg = int(args[args.index("--reps") + 1]) if "--reps" in args else 12
args = [a for i, a in enumerate(args) if a != "--reps" and (i == 0 or args[i - 1] != "--reps")]
if len(args) > 0:
    node = int(args[0])
    cpus = set(nodes[node]) & set(allowed)
    os.sched_setaffinity(0, cpus)
    print("bound to node", node, "(%d cpus)" % len(cpus))
import proof_systems_amd.khip as khip  # noqa: E402

khip.init(0)
khip.set_phase_timers(False)
rng = np.random.default_rng(3)
log_n = 16
co = rng.integers(0, 1 << 62, size=(16, 1 << log_n, 4), dtype=np.uint64)
outs = [np.ones((1, 8 << log_n, 4), np.uint64) for _ in range(16)]
if register:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    t0 = time.perf_counter()
    for a in [co] + outs:
        rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), a.nbytes, 0)
        assert rc == 0, "hipHostRegister failed: %d" % rc
    print("registered %d MB of caller buffers in %.1f ms" % ((co.nbytes + sum(o.nbytes for o in outs)) >> 20, 1e3 * (time.perf_counter() - t0)))
fs = [(lambda i=i: khip.lde(khip.FP, co[i:i + 1], log_n, 3, out=outs[i])) for i in range(16)]
bar = threading.Barrier(17); done = threading.Barrier(17)
reps = reps_arg


def w(f):
    for _ in range(reps):
        bar.wait(); f(); done.wait()


th = [threading.Thread(target=w, args=(f,)) for f in fs]
for t in th:
    t.start()
ts = []
for _ in range(reps):
    bar.wait(); t0 = time.perf_counter(); done.wait(); ts.append(time.perf_counter() - t0)
for t in th:
    t.join()
srt = sorted(ts[3:])
print("16 extensions 2^16 -> 2^19 from 16 threads, %s host buffers: best %.2f ms, median %.2f, worst %.2f, over 8 ms: %d of %d; all: %s"
      % ("REGISTERED" if register else "pageable", 1e3 * srt[0], 1e3 * srt[len(srt) // 2], 1e3 * srt[-1], sum(1 for t in srt if t > 8e-3), len(srt), " ".join("%.1f" % (1e3 * t) for t in ts)))
