"""Host-only entry points of the library the transcript and the index need -- no device involved, so they are compared with the oracle here on the
CPU: ScalarChallenge::to_field (poseidon/src/sponge.rs:190-226), the group map of SRS::open's u (groupmap/src/lib.rs:74-188), the evaluation
domains' generators (ark-poly Radix2EvaluationDomain::new), SRS::create's points and blinding base (poly-commitment/src/ipa.rs:751-778)."""
import random

import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    return k


def _val(F, limbs):
    return F.from_mont(P.from_limbs([int(x) for x in limbs]))


def _limbs(F, v):
    return np.array(P.to_limbs(F.to_mont(v % F.p)), dtype=np.uint64)


@pytest.mark.parametrize("cid", [0, 1])
def test_scalar_challenge_to_field(khip, cid):
    C = P.CURVES[cid]
    _, er = P.endos(C)
    rnd = random.Random(5 + cid)
    for chal in [0, 1, 2, 3, (1 << 128) - 1, 1 << 127, 0x5555_5555_5555_5555_5555_5555_5555_5555] + [rnd.getrandbits(128) for _ in range(40)]:
        assert _val(C.scalar, khip.scalar_challenge_to_field(cid, chal)) == P.challenge_to_field(C.scalar, chal, er), hex(chal)


@pytest.mark.parametrize("cid", [0, 1])
def test_group_map_to_group(khip, cid):
    C = P.CURVES[cid]
    F = C.base
    rnd = random.Random(9 + cid)
    for t in [0, 1, 2, F.p - 1] + [rnd.randrange(F.p) for _ in range(24)]:
        got = khip.group_map_to_group(cid, _limbs(F, t))
        x, y = C.to_group(t)
        assert (_val(F, got[:4]), _val(F, got[4:])) == (x, y), t
        assert (y * y - x * x * x - C.b) % F.p == 0


@pytest.mark.parametrize("fid", [0, 1])
def test_domain_generators(khip, fid):
    F = P.Fp if fid == 0 else P.Fq
    for log2_n in range(0, 33):
        w = _val(F, khip.domain_generator(fid, log2_n))
        assert w == F.root_of_unity(log2_n)
        assert pow(w, 1 << log2_n, F.p) == 1 and (log2_n == 0 or pow(w, 1 << (log2_n - 1), F.p) == F.p - 1)


@pytest.mark.parametrize("cid", [0, 1])
def test_srs_points_and_blinding_base(khip, cid):
    """the library's host generator = the oracle's C port (itself pinned on the reference's srs files and literals, tests/test_oracle_kats.py), at
    the start, across a block boundary of the thread split and far out"""
    for start, count in ((0, 70), (65530, 12), (1 << 20, 5)):
        assert np.array_equal(khip.srs_generate(cid, start, count, threads=3), cref.srs_generate(cid, start, count, threads=2))
    assert np.array_equal(khip.srs_h(cid), cref.srs_h(cid))
