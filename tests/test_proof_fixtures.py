"""The committed whole-proof fixtures (tests/golden/proof_fixtures/, made by tests/golden/make_proof_fixtures.py with the oracle's CPU prover):
records are consistent with their bytes, parse as the reference's `ProverProof` layout, and the small case is re-derived from scratch here
(circuit + seed -> the same bytes, with the transforms in the C oracle and on Python integers: the generator is deterministic and the
committed files are its output).  The 2^16 cases take minutes each to re-derive and are not re-run by the suite; the generator asserted
that the oracle verifier accepts each proof (`accepted_by_oracle_verifier`).  CPU only."""
import hashlib
import json
import os
import sys

import pytest

from oracle import pasta as P
from oracle import prover as OPR

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "proof_fixtures")
sys.path.insert(0, os.path.join(HERE, "golden"))
NAMES = ["bench_vesta_2_10", "bench_vesta_2_16", "bench_pallas_2_16", "bench_vesta_2_17_over_2_16", "bench_vesta_2_16_prev1", "and_lookup_vesta_2_13", "generic_public_vesta_2_16", "library_gates_vesta_2_13"]


def _load(name):
    with open(os.path.join(FIX, name + ".json")) as f:
        rec = json.load(f)
    with open(os.path.join(FIX, name + ".proof.bin"), "rb") as f:
        return rec, f.read()


@pytest.mark.parametrize("name", NAMES)
def test_record_matches_bytes_and_layout(name):
    import msgpack
    rec, raw = _load(name)
    assert len(raw) == rec["proof_len"] and hashlib.sha256(raw).hexdigest() == rec["proof_sha256"]
    assert rec["accepted_by_oracle_verifier"] is True
    commitments, opening, evals, ft_eval1, prev = msgpack.unpackb(raw, raw=True)
    nch = rec["num_chunks"]
    assert len(commitments[0]) == 15 and all(len(c[0]) == nch for c in commitments[0]) and len(commitments[2][0]) == 7 * nch
    assert (commitments[3] is not None) == bool(rec.get("lookup")) and len(prev) == rec.get("prev_challenges", 0)
    assert len(opening[0]) == rec["log2_srs"] and len(ft_eval1) == 32
    assert len(evals[1]) == 15 and all(len(e[0]) == nch and len(e[1]) == nch for e in evals[1])


def test_small_fixture_is_what_the_generator_writes_today():
    import make_proof_fixtures as M
    rec, raw = _load("bench_vesta_2_10")
    cid, log2_n, log_srs, seed = M.CASES["bench_vesta_2_10"]
    C = P.CURVES[cid]
    cs, rows = M.bench_circuit(C.scalar, log2_n, log_srs)
    ix = OPR.Index(C, cs, OPR.Srs(C, 1 << log_srs))
    assert hex(ix.digest) == rec["verifier_index_digest_hex"]
    for fast in (8, None):                                     # transforms in the C oracle / on Python integers: the same bytes
        OPR.FAST_LOG = fast
        try:
            proof = OPR.create_proof(ix, [[1] * rows for _ in range(15)], P.StdRng(seed))
        finally:
            OPR.FAST_LOG = 8
        assert OPR.serialize_proof(C, proof) == raw
