"""KH_NTT29=1: the NTT passes on nine 29-bit limbs (csrc/ntt29.cuh, k_ntt_pass29) -- the measured-and-not-default variant of round 4 -- must stay
bit-exact: the library's own 32-bit-limb passes are the reference here (they are held to the oracle by tests/test_gpu_parity.py), over every
pass shape the driver produces (1-3 passes, odd and even stage counts, batches, both fields, inverse with the 1/N in the inter-pass table,
the extension with its virtual first digit, the coset transform).  The switch is read once per process, so each side runs in its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1])
import proof_systems_amd.khip as khip
khip.init(0)
rng = np.random.default_rng(2929)
h = hashlib.sha256()
def rs(m):
    s = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 61) - 1); return s
for field in (khip.FP, khip.FQ):
    for logn, batch in ((1, 3), (2, 5), (3, 1), (5, 7), (8, 3), (9, 2), (10, 19), (13, 2), (16, 3), (17, 1), (19, 1)):
        for inv in (False, True):
            x = rs(batch << logn)
            h.update(khip.ntt(field, x.reshape(batch, 1 << logn, 4), logn, inverse=inv).tobytes())
    for logn, logb, batch in ((4, 3, 2), (8, 3, 5), (12, 3, 3), (16, 3, 2), (10, 2, 4), (9, 1, 1)):
        h.update(khip.lde(field, rs(batch << logn).reshape(batch, 1 << logn, 4), logn, logb).tobytes())
edge = np.zeros((1 << 10, 4), dtype=np.uint64)                      # zeros, one, p - 1 (as limbs of valid Montgomery values)
edge[1] = [0xfffffffd, 0, 0, 0]; edge[2] = [0x992d30ed00000000, 0x224698fc094cf91b, 0, 0x4000000000000000]
h.update(khip.ntt(khip.FP, edge.reshape(1, 1 << 10, 4), 10, inverse=False).tobytes())
print("NTT_DIGEST", h.hexdigest())
"""


def _digest(env_val):
    env = dict(os.environ, KH_NTT29=env_val)
    p = subprocess.run([sys.executable, "-c", WORKER, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return [l for l in p.stdout.decode().splitlines() if l.startswith("NTT_DIGEST")][0]


def test_29_bit_passes_equal_the_32_bit_passes():
    assert _digest("1") == _digest("0")
