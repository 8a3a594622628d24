#!/usr/bin/env python3
"""Digest of 200,000 chained Poseidon permutations per run (both fields, random transcripts salted with boundary values: 0, 1, p - 1, limb boundaries of the
52-bit form ...).  Run it twice -- plain and with KH_NO_IFMA=1 -- and compare: the AVX-512 IFMA and the scalar permutation must print the same digest.
Usage: poseidon_stress.py SEED    (no GPU needed; round 5: seeds 1-3 equal, 600,000 permutations)"""
import sys, hashlib, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import proof_systems_amd.khip as khip
from oracle import pasta as P
rng = np.random.default_rng(int(sys.argv[1]))
h = hashlib.sha256()
for kind, curve, F in ((khip.Sponge.FR, khip.VESTA, P.Fp), (khip.Sponge.FR, khip.PALLAS, P.Fq)):
    p = F.p
    specials = [0, 1, 2, p - 1, p - 2, (1 << 254) - 1, (1 << 254), (1 << 208) - 1, (1 << 208), (1 << 52) - 1, (1 << 104) - 1, (1 << 156) - 1, p >> 1, (p >> 1) + 1,
                0x3fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff % p, int("f" * 63, 16) % p]
    sp_l = np.array([P.to_limbs(v % p) for v in specials], dtype=np.uint64)
    for rep in range(200):
        sp = khip.Sponge(kind, curve)
        x = rng.integers(0, 1 << 64, size=(1000, 4), dtype=np.uint64)
        x[:, 3] &= np.uint64((1 << 62) - 1)            # < 2^254 < p
        idx = rng.integers(0, 1000, size=64)
        x[idx] = sp_l[rng.integers(0, len(specials), size=64)]
        sp.absorb(x)
        h.update(sp.squeeze_field().tobytes()); h.update(sp.digest().tobytes())
print(h.hexdigest())
