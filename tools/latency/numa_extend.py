#!/usr/bin/env python3
"""Is the host-buffer path's slow mode (16 concurrent eightfold extensions: 5.5 ms in some processes, 11 in others) a NUMA placement effect?  Runs the same
16 threaded kh_lde calls with the process (and therefore the first touch of its pageable buffers) bound to the CPUs of each NUMA node in turn.
Usage: numa_extend.py            (prints the topology, the GPU's node, and the time per node)"""
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
if len(sys.argv) > 1:
    node = int(sys.argv[1])
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
fs = [(lambda i=i: khip.lde(khip.FP, co[i:i + 1], log_n, 3, out=outs[i])) for i in range(16)]
bar = threading.Barrier(17); done = threading.Barrier(17)
reps = 12


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
print("16 extensions 2^16 -> 2^19 from 16 threads, host buffers: best %.2f ms, all: %s" % (1e3 * min(ts[3:]), " ".join("%.1f" % (1e3 * t) for t in ts)))
