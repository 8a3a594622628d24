"""Round-6 entry points on the device against the C oracle: kh_msm_submit_host (the MSM pipeline from HOST scalars -- what
SRS::commit_non_hiding(&DensePolynomial) hands over, poly-commitment/src/ipa.rs:638-683) in its chunked and un-chunked forms, and the lone big host MSM
(the chunked upload -- KH_HOST_CHUNK_MIN -- and the two half-range jobs of a lone big MSM -- KH_HOST_SPLIT_MIN -- are
experiments that measured level and are off by default: the tests below run them in a subprocess)."""
import os

import numpy as np
import pytest

from oracle import cref

pytestmark = pytest.mark.gpu
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


@pytest.mark.parametrize("cid", [0, 1])
def test_msm_submit_host_pipeline(khip, cid):
    """Three MSMs in flight from host buffers that are OVERWRITTEN as soon as the submit returns (the contract: the scalars are the caller's again), sizes on
    both sides of the chunking threshold (2^17 scalars), ragged lengths, an offset window, canonical and Montgomery scalars, and a batch of two (the
    un-chunked asynchronous upload); every result against the C oracle."""
    rng = np.random.default_rng(600 + cid)
    N = 1 << 18
    g = cref.srs_generate(cid, 0, N, threads=THREADS)
    srs = khip.Srs(cid, g)
    cases = [(1 << 18, 0, True), (1 << 17, 1 << 17, False), ((1 << 17) + 12345, 777, True), (5000, 3, True), (1 << 16, 0, False), ((1 << 18) - 1, 1, True)]
    inflight, want = [], []
    for n, off, mont in cases * 2:
        sc = _rand_fe(rng, n)
        want.append(cref.msm(cid, g[off:off + n], sc, scalars_mont=mont, threads=THREADS))
        buf = sc.copy()
        inflight.append(srs.msm_submit_host(buf, offset=off, mont=mont))
        buf[:] = 0xFFFFFFFFFFFFFFFF                       # the caller reuses its buffer at once
        if len(inflight) == 3:
            out, inf = khip.Srs.msm_wait(inflight.pop(0))
            w, winf = want.pop(0)
            assert bool(inf[0]) == bool(winf) and np.array_equal(out[0], w)
    while inflight:
        out, inf = khip.Srs.msm_wait(inflight.pop(0))
        w, winf = want.pop(0)
        assert bool(inf[0]) == bool(winf) and np.array_equal(out[0], w)
    # a batch of two from host memory
    n = 1 << 14
    sc2 = _rand_fe(rng, 2 * n).reshape(2, n, 4)
    t = srs.msm_submit_host(sc2.copy(), k=2)
    out, inf = khip.Srs.msm_wait(t)
    for j in range(2):
        w, winf = cref.msm(cid, g[:n], sc2[j], threads=THREADS)
        assert not inf[j] and np.array_equal(out[j], w)
    # zero scalars: the identity
    out, inf = khip.Srs.msm_wait(srs.msm_submit_host(np.zeros((1 << 17, 4), np.uint64)))
    assert inf[0]
    srs.close()


def test_msm_submit_host_from_pinned_memory(khip):
    """Pinned host memory makes the upload truly asynchronous: the call must still not return before the scalars have been read."""
    import torch
    rng = np.random.default_rng(611)
    n = 1 << 18
    g = cref.srs_generate(0, 0, n, threads=THREADS)
    srs = khip.Srs(0, g)
    sc = _rand_fe(rng, n)
    want, winf = cref.msm(0, g, sc, threads=THREADS)
    pinned = torch.empty((n, 4), dtype=torch.int64).pin_memory()
    view = pinned.numpy().view(np.uint64)
    for _ in range(4):
        view[:] = sc
        t = srs.msm_submit_host(view)
        view[:] = 0
        out, inf = khip.Srs.msm_wait(t)
        assert not inf[0] and np.array_equal(out[0], want)
    srs.close()


def test_chunked_upload_and_split_experiments_stay_bit_exact():
    """KH_HOST_CHUNK_MIN / KH_HOST_SPLIT_MIN (off by default) in a fresh process: the same MSMs against the C oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import cref\n"
        "import proof_systems_amd.khip as k\n"
        "k.init(0)\n"
        "rng = np.random.default_rng(5)\n"
        "n = 1 << 17\n"
        "g = cref.srs_generate(0, 0, n, threads=8)\n"
        "srs = k.Srs(0, g)\n"
        "for m in (n, n - 77, 1 << 16):\n"
        "    sc = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 61) - 1)\n"
        "    w, winf = cref.msm(0, g[:m], sc, threads=8)\n"
        "    o, i = srs.msm(sc)\n"
        "    assert not i and np.array_equal(o, w), ('sync', m)\n"
        "    o, i = k.Srs.msm_wait(srs.msm_submit_host(sc))\n"
        "    assert not i[0] and np.array_equal(o[0], w), ('submit_host', m)\n"
        "print('ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KH_HOST_CHUNK_MIN="4096", KH_HOST_SPLIT_MIN="65536"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode()[-1500:]
