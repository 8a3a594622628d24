"""The library's host-side Fiat-Shamir sponges (csrc/host_sponge.cpp, kh_sponge_*) against the oracle's restatement
(oracle/poseidon.py, itself pinned on the reference's poseidon/tests/test_vectors/kimchi.json and on the opening-proof
bytes of poly-commitment/tests/commitment.rs:388-440).  Host code: runs without a GPU."""
import json
import os
import random

import numpy as np
import pytest

from oracle import pasta as P
from oracle import poseidon as OP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    return k


def L(F, v):
    return np.array(P.to_limbs(F.to_mont(v % F.p)), dtype=np.uint64)


def V(F, limbs):
    return F.from_mont(P.from_limbs([int(x) for x in limbs]))


def test_reference_poseidon_vectors(khip):
    """poseidon/tests/test_vectors/kimchi.json through the library's Fr-sponge (absorb all inputs, squeeze one element)."""
    allk = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon_kimchi_params.json")))["kats"]
    kats = allk["kimchi_fp_hash"]
    assert len(kats) >= 5
    # the empty-transcript challenge regressions (poseidon/src/sponge.rs tests): FqSponge of Vesta / Pallas
    assert khip.Sponge(khip.Sponge.FQ, khip.VESTA).challenge() == int.from_bytes(bytes.fromhex(allk["challenge_empty_vesta"]), "little")
    assert khip.Sponge(khip.Sponge.FQ, khip.PALLAS).challenge() == int.from_bytes(bytes.fromhex(allk["challenge_empty_pallas"]), "little")
    for kat in kats:
        s = khip.Sponge(khip.Sponge.FR, khip.VESTA)                       # scalar field of Vesta = Fp, the vectors' field
        ins = [int.from_bytes(bytes.fromhex(h), "little") for h in kat["input"]]
        if ins:
            s.absorb(np.stack([L(P.Fp, v) for v in ins]))
        got = V(P.Fp, s.squeeze_field())
        assert got == int.from_bytes(bytes.fromhex(kat["output"]), "little")


@pytest.mark.parametrize("cid", [0, 1])
def test_fq_sponge_transcript_matches_oracle(khip, cid):
    """A random interleaving of every FqSponge operation the prover uses, library vs oracle, both curves (the absorb_fr
    encoding differs between them: sponge.rs:337-366)."""
    curve = P.CURVES[cid]
    rnd = random.Random(40 + cid)
    lib = khip.Sponge(khip.Sponge.FQ, cid)
    ora = OP.DefaultFqSponge(curve)
    for step in range(200):
        op = rnd.randrange(7)
        if op == 0:
            k = rnd.randrange(1, 4)
            pts = [curve.mul(curve.gen, rnd.randrange(1, 1 << 40)) if rnd.random() < 0.85 else None for _ in range(k)]
            xy = np.zeros((k, 8), np.uint64); inf = np.zeros(k, np.uint8)
            for i, pt in enumerate(pts):
                if pt is None:
                    inf[i] = 1
                else:
                    xy[i, :4] = L(curve.base, pt[0]); xy[i, 4:] = L(curve.base, pt[1])
            lib.absorb_g(xy, inf); ora.absorb_g(pts)
        elif op == 1:
            v = rnd.randrange(curve.base.p)
            lib.absorb(L(curve.base, v)); ora.absorb_fq([v])
        elif op == 2:
            v = rnd.choice([rnd.randrange(curve.scalar.p), curve.scalar.p - 1, 0, 1])
            lib.absorb_fr(L(curve.scalar, v)); ora.absorb_fr([v])
        elif op == 3:
            assert lib.challenge() == ora.challenge()
        elif op == 4:
            assert V(curve.scalar, lib.challenge_field()) == ora.challenge()
        elif op == 5:
            assert V(curve.base, lib.squeeze_field()) == ora.challenge_fq()
        else:
            c1, c2 = lib.clone(), ora.clone()                             # fq_sponge.clone() keeps the squeeze buffer
            assert c1.challenge() == c2.challenge()
            got = V(curve.scalar, c1.digest())
            x = c2.challenge_fq()
            assert got == (x if x < curve.scalar.p else 0)
