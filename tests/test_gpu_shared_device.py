"""The one-launch sort (k_sort_fused, csrc/msm.hip) spins on grid barriers and needs all its blocks resident at once.  Inside one process
that is guaranteed; with several PROCESSES on one GPU it is not.  Its barriers are bounded: after KH_FUSED_SPIN_US they give up, the
host re-runs the job with the multi-launch sort and the process stays on it.  Here: (1) the fallback forced on every job (spin limit 0)
gives bit-exact results -- MSM pairs against the C oracle and the reference's opening-proof bytes; (2) eight processes with four small
MSM pairs in flight each on ONE GPU all finish, with correct results, inside a time bound (no spin deadlock)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, threading
import numpy as np
sys.path.insert(0, %r)
import proof_systems_amd.khip as khip
from oracle import cref
khip.init(0)
n = 1 << 14
srs = khip.Srs.create(khip.VESTA, n)
g = srs.get_g()
rng = np.random.default_rng(int(sys.argv[1]))
sc = rng.integers(0, 1 << 64, size=(2, n, 4), dtype=np.uint64); sc[:, :, 3] &= np.uint64((1 << 61) - 1)
want = [cref.msm(0, g, sc[j], threads=4)[0] for j in range(2)]
d = khip.DevBuf(sc.nbytes).upload(sc)
bad = []
def run():
    for _ in range(int(sys.argv[2])):
        out, inf = srs.msm_batch_dev(d.ptr, n, 2)          # k = 2 over the tables: the fused one-launch sort (an opening round's L / R shape)
        if not (np.array_equal(out[0], want[0]) and np.array_equal(out[1], want[1])): bad.append(1)
th = [threading.Thread(target=run) for _ in range(4)]
for t in th: t.start()
for t in th: t.join()
print("OK" if not bad else "MISMATCH", len(bad))
''' % ROOT


def test_forced_fallback_is_bit_exact():
    env = dict(os.environ, KH_FUSED_SPIN_US="0")
    r = subprocess.run([sys.executable, "-c", WORKER, "1", "5"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    assert "multi-launch sort from now on" in r.stderr, "the fallback did not trigger with a zero spin limit"
    # the opening rounds (graph replay of the same launch sequence) under the forced fallback: the reference's opening-proof bytes
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(ROOT, "tests", "test_gpu_open_kat.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])


def test_eight_processes_four_in_flight_on_one_gpu():
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(10 + i), "40"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(8)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a process did not finish: spin deadlock on the shared GPU")
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, (o[-300:], e[-800:])
