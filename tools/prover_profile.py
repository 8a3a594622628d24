"""cProfile of proof_systems_amd.prover.create_proof (benchmark circuit, 2^16): where the host side of a proof spends its time.
This is how the 3.8 ms per proof of hipFree calls were found (now pooled).  Usage: python tools/prover_profile.py"""
import os, sys, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proof_systems_amd.khip as khip
from proof_systems_amd import prover
khip.init(0)
ix = prover.bench_circuit_index(khip.VESTA, 16)
F = prover.Fld(ix.fid)
wit = np.tile(F.limbs(1), (15, (1 << 16) - 10, 1))
rng = np.random.default_rng(1)
for _ in range(3): prover.create_proof(ix, wit, rng, check=False)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): prover.create_proof(ix, wit, rng, check=False)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
