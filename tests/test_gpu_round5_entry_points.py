"""The small entry points round 5 added to the C ABI, on the device: kh_dev_fill_elements (32-byte records set from a value that travels in the kernel's
arguments; what kh_prove uses for its one-row patches), the small-size path of kh_dev_memset_zero, small argument-table uploads carried by a kernel
(exercised through an expression with constants and through kh_poly_lincomb_dev by the other suites; here: switched off and on, same results), and
kh_set_phase_timers (per-phase HIP events behind kh_last_timings: off by default in the library, on after khip.init)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def test_fill_elements_and_small_memset(khip):
    rng = np.random.default_rng(5)
    n = 1000
    base = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    buf = khip.DevBuf(n * 32).upload(base)
    v1 = rng.integers(0, 1 << 63, size=4, dtype=np.uint64)
    v2 = rng.integers(0, 1 << 63, size=4, dtype=np.uint64)
    buf.fill_elements(3, v1, 1)                        # one row, as the prover's patches
    buf.fill_elements(100, v2, 777)                    # several blocks
    got = buf.download((n, 4))
    want = base.copy(); want[3] = v1; want[100:877] = v2
    assert np.array_equal(got, want)
    khip.dev_memset_zero(buf.ptr + 32 * 10, 32 * 5)    # small: the kernel path
    want[10:15] = 0
    assert np.array_equal(buf.download((n, 4)), want)
    buf.zero()                                         # 32,000 bytes: still the kernel path
    assert not buf.download((n, 4)).any()
    big = khip.DevBuf(1 << 20).upload(np.ones(1 << 17, np.uint64))
    big.zero()                                         # 1 MiB: the memset path
    assert not big.download((1 << 17,)).any()
    buf.free(); big.free()


def test_phase_timers_switch(khip):
    n = 1 << 12
    rng = np.random.default_rng(6)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 61) - 1)
    srs = khip.Srs.create(khip.VESTA, n)
    a, _ = srs.msm(sc)
    assert any(name == "accumulate" for name, _ms in khip.last_timings())          # khip.init switched them on
    khip.set_phase_timers(False)
    b, _ = srs.msm(sc)
    assert khip.last_timings() == []
    khip.set_phase_timers(True)
    c, _ = srs.msm(sc)
    assert any(name == "accumulate" for name, _ms in khip.last_timings())
    srs.close()
    assert np.array_equal(a, b) and np.array_equal(a, c)


WORKER = r"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1])
import proof_systems_amd.khip as khip
from proof_systems_amd import prover
khip.init(0)
ix = prover.bench_circuit_index(khip.VESTA, 10)
wit = np.tile(ix.F.limbs(1), (15, (1 << 10) - 10, 1))
proof = prover.create_proof_native(ix, wit, np.random.default_rng(3))
from oracle import views as V
print(hashlib.sha256(repr(V.device_views(ix, proof)[2]).encode()).hexdigest())
"""


def test_argument_uploads_in_kernel_arguments_change_nothing():
    """KH_STAGE_PUT=0 (asynchronous copies from the pinned ring, round 4's way) and the default (a kernel whose argument block carries the table) give the
    same proof, seeded: every token program, pointer table and constant block of a proof goes through one or the other."""
    outs = []
    for put in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", WORKER, ROOT], env=dict(os.environ, KH_STAGE_PUT=put), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        outs.append(r.stdout.decode().strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 64


XFER_WORKER = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import proof_systems_amd.khip as khip
from oracle import cref
khip.init(0)
FP = khip.FP
rng = np.random.default_rng(11)
for rep in range(3):                                   # every call allocates afresh (the bound is 1 MB, the buffers are 2.3 / 18 MB) and gives them back
    x = rng.integers(0, 1 << 62, size=(4, 1 << 14, 4), dtype=np.uint64)
    ev = khip.ntt(FP, x, 14, False)
    assert np.array_equal(ev, cref.ntt(FP, x, 14, False)), ("ntt", rep)
    back = khip.ntt(FP, ev, 14, True)
    assert np.array_equal(back, x), ("round trip", rep)
    e8 = khip.lde(FP, x, 14, 3)
    assert np.array_equal(e8, cref.lde(FP, x, 14, 3)), ("lde", rep)
print("transforms ok")
"""


def test_host_buffer_transforms_release_oversized_thread_buffers():
    """kh_ntt / kh_lde on host buffers keep two device buffers per calling thread -- up to a bound (1 GB each; KH_XFER_KEEP_MB overrides), above which a
    call gives them back when it ends.  With the bound at 1 MB every call takes that path: same results, call after call."""
    r = subprocess.run([sys.executable, "-c", XFER_WORKER, ROOT], env=dict(os.environ, KH_XFER_KEEP_MB="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert r.stdout.decode().strip().endswith("transforms ok")


FLAG_WORKER = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import proof_systems_amd.khip as khip
from oracle import cref
khip.init(0)
khip.set_phase_timers(False)                         # the library's default: completion by flag where the last kernel stores it
rng = np.random.default_rng(21)
n = 1 << 13
srs = khip.Srs.create(khip.VESTA, n)
g = srs.get_g()
for rep in range(6):
    k = 1 + rep % 3
    sc = rng.integers(0, 1 << 64, size=(k, n, 4), dtype=np.uint64); sc[:, :, 3] &= np.uint64((1 << 61) - 1)
    d = khip.DevBuf(sc.nbytes).upload(sc)
    xy, inf = srs.msm_batch_dev(d.ptr, n, k)
    d.free()
    for j in range(k):
        want, winf = cref.msm(khip.VESTA, g, sc[j], scalars_mont=True, threads=2)
        assert bool(inf[j]) == bool(winf) and (winf or np.array_equal(np.asarray(xy[j]).reshape(8), want)), (rep, j)
srs.close()
print("msm ok")
"""


@pytest.mark.parametrize("env", [{}, {"KH_NO_DONE_FLAG": "1"}, {"KH_NO_FIN_QUAD": "1"}])
def test_completion_flag_and_its_fallbacks(env):
    """Synchronous MSMs with the completion word (default), without it (KH_NO_DONE_FLAG: the event), and with a last kernel that does NOT store it
    (KH_NO_FIN_QUAD: the host's launch count runs ahead, every wait falls back to the event and resyncs): the same results against the oracle."""
    r = subprocess.run([sys.executable, "-c", FLAG_WORKER, ROOT], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert r.stdout.decode().strip().endswith("msm ok")
