"""The opening's rebase (csrc/rebase.hip): the folded basis of SRS::open after j0 rounds (poly-commitment/src/ipa.rs:985-1003 folds once per round;
g'[i] = sum_q coef[q] G[q N + i] with coef = the tensor of (1, u_k) is the same basis after j0 folds) materialised from the c = 16 window tables, against
the C oracle's MSM per output; then whole openings that switch to the materialised basis at the earliest round (KH_IPA_REBASE_WAIT=1), against openings
that never do (KH_IPA_REBASE=0) -- L, R of every round, a0, b0, sg must be the same bytes."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THREADS = min(64, os.cpu_count() or 8)


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def _rand_fe(rng, n):
    c = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    c[:, 3] &= np.uint64((1 << 61) - 1)
    return c


@pytest.mark.parametrize("cid,logn,logq", [(0, 10, 3), (1, 10, 4), (0, 12, 6), (0, 13, 1), (1, 11, 5)])
def test_rebase_points_match_the_oracle(khip, cid, logn, logq):
    """every output against an oracle MSM over the strided points; scalars: random, plus the digit patterns at the edges of the (hi, lo) bucket split
    (|d| = 2^15: hi = 128, lo = 0; d = -1; d = 255 / 256 / 257; zero scalars)"""
    rng = np.random.default_rng(100 * logn + logq + cid)
    n, Q = 1 << logn, 1 << logq
    N = n // Q
    F = P.CURVES[cid].scalar
    g = cref.srs_generate(cid, 0, n, threads=THREADS)
    srs = khip.Srs(cid, g)
    coef = _rand_fe(rng, Q)
    special = [0, 1, F.p - 1, 0x8000, 0x8000 << 16, 255, 256, 257, (1 << 255) % F.p, F.p - 0x8000, 0x00ff00ff00ff00ff, 0x80008000800080008000]
    mont = cref.ints_to_limbs([F.to_mont(v % F.p) for v in special])
    for j in range(min(Q, len(special))):
        coef[Q - 1 - j] = mont[j]
    if Q >= 2:
        coef[0] = mont[1]                                  # coef[0] = 1 as in a challenge tensor
    got = srs.debug_rebase_points(coef)
    for i in sorted(set(list(range(0, N, max(1, N // 40))) + [N - 1, 1, 63, min(64, N - 1), N // 2 + 17])):
        want, winf = cref.msm(cid, g[i::N], coef, threads=4)
        assert not winf and np.array_equal(got[i], want), (i,)
    srs.close()


def test_rebase_reports_an_output_at_infinity(khip):
    """all-zero scalars: every output is the point at infinity, which has no affine table entry -- the materialisation must say so (the opening then stays
    on the original basis: counter rebase_abandon)"""
    g = cref.srs_generate(0, 0, 1 << 10, threads=THREADS)
    srs = khip.Srs(0, g)
    with pytest.raises(Exception, match="infinity"):
        srs.debug_rebase_points(np.zeros((8, 4), np.uint64))
    srs.close()


WORKER = r"""
import sys, hashlib, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import proof_systems_amd.khip as khip
khip.init(0)
out = {}
for cid, logn in ((0, 10), (1, 12), (0, 13)):
    n = 1 << logn
    rng = np.random.default_rng(31 * logn + cid)
    def rs(k):
        a = rng.integers(0, 1 << 64, size=(k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
    srs = khip.Srs.create(cid, n)
    U = khip.srs_generate(cid, 1 << 21, 1)[0]
    a = rs(n - 5); b = rs(n); r = rs(2 * logn)
    chals = [int.from_bytes(rng.bytes(16), "little") for _ in range(logn)]
    h = hashlib.sha256()
    for rep in range(2):
        op = khip.IpaOpening(srs, a, b, U)
        for j, ch in enumerate(chals):
            xy, inf = op.round_lr(r[2 * j], r[2 * j + 1]); h.update(xy.tobytes()); h.update(inf.tobytes())
            u, ui = op.round_fold(ch); h.update(u.tobytes())
        a0, b0, sg, sginf = op.finish(); h.update(a0.tobytes()); h.update(b0.tobytes()); h.update(np.asarray(sg).tobytes()); h.update(bytes([int(bool(sginf))]))
        op.free()
    out["%d_%d" % (cid, logn)] = h.hexdigest()
out["rebase_switch"] = khip.counter("rebase_switch"); out["rebase_launch"] = khip.counter("rebase_launch"); out["rebased_rounds"] = khip.counter("rebased_rounds")
out["rebase_abandon"] = khip.counter("rebase_abandon")
print(json.dumps(out))
"""


def _run(env):
    import json
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def test_openings_are_the_same_bytes_with_and_without_the_rebase(khip):
    """Three SRS sizes, both curves, two openings each (the handle's rebase workspaces are reused): never rebased / switched at the earliest possible round
    / switched whenever the background job happens to be ready / a materialised basis of 64 points with 8-bit windows."""
    off = _run({"KH_IPA_REBASE": "0"})
    assert off["rebase_launch"] == 0 and off["rebased_rounds"] == 0
    early = _run({"KH_IPA_REBASE_WAIT": "1"})
    assert early["rebase_launch"] == 6 and early["rebase_switch"] == 6 and early["rebase_abandon"] == 0 and early["rebased_rounds"] > 0
    free = _run({})
    small = _run({"KH_IPA_REBASE_WAIT": "1", "KH_IPA_REBASE_LOGN": "6", "KH_IPA_REBASE_C": "8"})
    assert small["rebase_switch"] == 6
    for k in ("0_10", "1_12", "0_13"):
        assert early[k] == off[k], ("early switch", k)
        assert free[k] == off[k], ("free-running switch", k)
        assert small[k] == off[k], ("64-point basis, c = 8", k)


@pytest.mark.parametrize("cid", [0, 1])
def test_glv_split_of_the_folded_basis_msms(cid):
    """The folded basis's window tables are built for 128-bit half-scalars (half the doubling chain) and the digit pass splits every scalar k = k1 + k2 lambda
    (csrc/msm.hip glv_split, constants from tools/gen_glv_params.py): the identity modulo r and the size bound, on the device, for edge values and 200,000 random
    scalars of both fields; lambda is the curve's endo_r (oracle.pasta.endos = poly-commitment/src/ipa.rs:214-231)."""
    import proof_systems_amd.khip as khip
    from oracle import cref
    from oracle import pasta as P
    khip.init(0)
    curve = P.CURVES[cid]
    r = curve.scalar.p
    _, lam = P.endos(curve)
    rng = np.random.default_rng(11 + cid)
    vals = [0, 1, 2, r - 1, r - 2, lam, r - lam, (r - 1) // 2, (r + 1) // 2] + [1 << i for i in range(255)] + [(1 << i) - 1 for i in range(1, 255)]
    vals = [v % r for v in vals]
    rnd = rng.integers(0, 1 << 64, size=(200000, 4), dtype=np.uint64)
    rnd[:, 3] &= np.uint64((1 << 62) - 1)
    rnd_int = [int(a) | int(b) << 64 | int(c) << 128 | int(d) << 192 for a, b, c, d in rnd[:2000]]      # (2000 of them checked in Python integers, all of them for size)
    rnd_int = [v % r for v in rnd_int]
    sc = cref.ints_to_limbs(vals + rnd_int)
    field = khip.FP if r == P.Fp.p else khip.FQ
    got = khip.glv_split(field, sc)
    for k, (k1, k2) in zip(vals + rnd_int, got):
        assert (k1 + k2 * lam - k) % r == 0, hex(k)
        assert abs(k1) < 1 << 127 and abs(k2) < 1 << 127, (hex(k), k1.bit_length(), k2.bit_length())
    # the bulk: canonical inputs below 2^254 < r
    big = khip.glv_split(field, rnd)
    assert max(max(abs(a), abs(b)) for a, b in big) < 1 << 127
