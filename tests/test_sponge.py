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


PERM_WORKER = r"""
import sys, hashlib
import numpy as np
sys.path.insert(0, sys.argv[1])
import proof_systems_amd.khip as khip
from oracle import pasta as P
rng = np.random.default_rng(2024)
h = hashlib.sha256()
for kind, curve in ((khip.Sponge.FR, khip.VESTA), (khip.Sponge.FR, khip.PALLAS)):       # the two fields' permutations
    sp = khip.Sponge(kind, curve)
    for rep in range(40):
        x = rng.integers(0, 1 << 64, size=(int(rng.integers(1, 9)), 4), dtype=np.uint64)
        x[:, 3] &= np.uint64((1 << 61) - 1)
        if rep % 7 == 0:
            x[0] = 0                                   # zero ...
        if rep % 5 == 0:                               # ... and the largest canonical limb pattern (p - 1 in the wire's Montgomery form is some value; as limbs: p - 1 itself is canonical too)
            x[-1] = np.array(P.to_limbs((P.Fp if curve == khip.VESTA else P.Fq).p - 1), dtype=np.uint64)
        sp.absorb(x)
        h.update(sp.squeeze_field().tobytes())
        h.update(sp.digest().tobytes())
print(h.hexdigest())
"""


def test_ifma_and_scalar_permutations_agree():
    """The AVX-512 IFMA permutation (three lanes, 52-bit limbs, R' = 2^260) and the scalar mulx one (KH_NO_IFMA=1) squeeze the same elements from
    the same random transcripts, both fields.  On a CPU without IFMA both runs take the scalar path (the test is then trivially true); the
    reference vectors above run on whichever path the CPU gives."""
    import subprocess
    import sys
    outs = []
    for no_ifma in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", PERM_WORKER, ROOT], env=dict(os.environ, KH_NO_IFMA=no_ifma), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        outs.append(r.stdout.decode().strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 64
