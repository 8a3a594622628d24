#!/usr/bin/env python3
"""Whole-proof fixtures at BASELINE config 3's OWN size (and config 5's Pallas half, and one chunked proof), made by the oracle's
CPU prover (oracle/prover.py -- pinned on the reference's seeded whole-proof regression, kimchi/src/tests/and.rs:404-731, by
tests/test_reference_kat.py; its transforms run in the C oracle from 2^8 points on, the SAME code path that reproduces those bytes).

Circuit = the reference's benchmark circuit, kimchi/src/bench.rs:59-122: 2^k - 10 generic gates `Const(1)`, every cell wired to
itself, a witness of ones, no public input.  The reference proves it with `OsRng`; a fixture needs a stream, so the draws come
from `StdRng::from_seed(seed)` (the oracle's restatement of rand 0.8.5 + ark-ff's `Fp::rand`, pinned by the commit / opening /
whole-proof vectors) -- the `rng: &mut RNG` a Rust caller would pass.  Written per case: the rmp-serde bytes of the `ProverProof`
(`<name>.proof.bin`, ~6 kB) and a JSON record (seed, sizes, sha256, verifier-index digest).  The `-m gpu` tests make `kh_prove` and the
Python device prover draw from the same stream and demand the same bytes (tests/test_gpu_proof_fixtures.py); a CPU test re-derives the
small case and checks the records (tests/test_proof_fixtures.py).

    python tests/golden/make_proof_fixtures.py [name ...]        # minutes per 2^16 case on 16 cores; nothing is read from /root/reference
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import circuit as CC  # noqa: E402
from oracle import kimchi as K  # noqa: E402
from oracle import pasta as P  # noqa: E402
from oracle import prover as OPR  # noqa: E402
from oracle import views as V  # noqa: E402

OUT = os.path.join(HERE, "proof_fixtures")

# name -> (curve id, log2 of the domain the circuit must land on, log2 of the SRS, seed)
CASES = {
    "bench_vesta_2_10": (0, 10, 10, bytes([16, 0] + [42] * 30)),          # small: re-derived by the CPU suite in seconds
    "bench_vesta_2_16": (0, 16, 16, bytes([16, 0] + [42] * 30)),          # BASELINE config 3
    "bench_pallas_2_16": (1, 16, 16, bytes([16, 1] + [42] * 30)),         # BASELINE config 5's other half
    "bench_vesta_2_17_over_2_16": (0, 17, 16, bytes([17, 0] + [42] * 30)),  # 2 chunks, zk_rows 5 (kimchi/src/tests/chunked.rs)
    # create_recursive at config 3's size: one previous challenge (16 folding challenges, its accumulator commitment), kimchi/src/tests/recursion.rs:44-75
    "bench_vesta_2_16_prev1": (0, 16, 16, bytes([16, 2] + [42] * 30)),
    # the lookup argument at size: AND gadgets (Xor16 rows with their 4-bit XOR-table lookups + generic rows, kimchi/src/tests/and.rs:126-160) filling a
    # 2^13 domain -- the circuit family of the reference's own whole-proof vector, 16 times its size
    "and_lookup_vesta_2_13": (0, 13, 13, bytes([13, 3] + [42] * 30)),
    # config 3's size with everything the benchmark circuit lacks: a RANDOM witness (uniform 62-bit values and their products: the fifteen witness
    # commitments are unskewed 2^16-point MSMs), addition + multiplication gates, three public inputs, copy constraints on every second row
    "generic_public_vesta_2_16": (0, 16, 16, bytes([16, 4] + [42] * 30)),
    # the gate library against BYTES: instances of Poseidon (11 rows = one 55-round permutation), CompleteAdd (incl. doubling, P + (-P)),
    # VarBaseMul, EndoMul, EndoMulScalar (witnesses by the reference's generators restated in oracle/gates.py) on a 2^13 domain (~120 instances each)
    "library_gates_vesta_2_13": (0, 13, 13, bytes([13, 5] + [42] * 30)),
}
PREV_SEED = bytes([11] * 32)           # the previous challenges of *_prev1 are drawn from StdRng::from_seed(PREV_SEED)
AND_SEED = bytes([12] * 32)            # the AND gadgets' 64-bit inputs


def bench_circuit(F, log2_n: int, log_srs: int):
    """BenchmarkCtx::new(log2_n) (bench.rs:59-96) as an oracle constraint system; over a shorter SRS the zero-knowledge rows grow
    (constraints.rs:769-771), so the gate count is what still fits: n - zk_rows - 7 (= 2^k - 10 for one chunk)."""
    p = F.p
    n = 1 << log2_n
    nch = 1 << max(0, log2_n - log_srs)
    zk = (16 * nch + 5) // 7
    rows = n - zk - 7
    gates = [CC.generic_gadget(p, r, CC.generic_spec(p, "Const", cst=1)) for r in range(rows)]
    cs = CC.build(F, gates, max_poly_size=(1 << log_srs) if log_srs < log2_n else None)
    assert cs["log2_n"] == log2_n and cs["zk_rows"] == zk, (cs["log2_n"], cs["zk_rows"])
    return cs, rows


def and_circuit(F, log2_n: int):
    """as many 8-byte AND gadgets (extend_and, and.rs:126-160) as fit the domain, with their witness (create_and_witness) on inputs from
    gen_field_with_bits(StdRng::from_seed(AND_SEED), 64)"""
    from oracle import gates as G
    p = F.p
    std = P.StdRng(AND_SEED)
    gates, rows = [], []
    per = None
    while per is None or len(gates) + per + 3 + 1 <= (1 << log2_n) - 1:
        before = len(gates)
        CC.extend_and(p, gates, 8)
        per = len(gates) - before
        rows += G.and_witness(F, CC.gen_field_with_bits(std, 64), CC.gen_field_with_bits(std, 64), 8)
    cs = CC.build(F, gates)
    assert cs["log2_n"] == log2_n and cs["lookup"] is not None and len(rows) == len(gates), (cs["log2_n"], len(rows), len(gates))
    CC.verify_witness(cs, [[r[c] for r in rows] for c in range(15)])
    return cs, rows


def generic_circuit(F, log2_n: int, log_srs: int, npub: int = 3, rng_seed: int = 20260926):
    """double generic gates a + b - c = 0 / a' b' - c' = 0 on random 62-bit operands, the first `npub` rows public inputs, (r, 0) ~ (r, 4) wired on every
    second row (generic.rs:380-470 style circuits; the shape tests/test_gpu_native_prover.py uses at small sizes)"""
    import numpy as np
    p = F.p
    rnd = np.random.default_rng(rng_seed)
    n = 1 << log2_n
    nch = 1 << max(0, log2_n - log_srs)
    zk = (16 * nch + 5) // 7
    rows = n - zk - 5
    ab = rnd.integers(1, 1 << 62, size=(rows, 2), dtype=np.int64)
    gates, wit = [], [[0] * rows for _ in range(15)]
    pub_spec, add_spec, mul_spec = CC.generic_spec(p, "Pub"), CC.generic_spec(p, "Add"), CC.generic_spec(p, "Mul")
    for r in range(rows):
        a, b = int(ab[r, 0]), int(ab[r, 1])
        if r < npub:
            gates.append(CC.generic_gadget(p, r, pub_spec))
            wit[0][r] = a
        else:
            gates.append(CC.generic_gadget(p, r, add_spec, mul_spec))
            wit[0][r], wit[1][r], wit[2][r] = a, b, (a + b) % p
            wit[3][r], wit[4][r], wit[5][r] = b, a, a * b % p
    for r in range(npub, rows - 1, 2):
        CC.connect_cell_pair(gates, (r, 0), (r, 4))
    cs = CC.build(F, gates, public=npub, max_poly_size=(1 << log_srs) if log_srs < log2_n else None)
    assert cs["log2_n"] == log2_n and cs["zk_rows"] == zk
    return cs, wit


def library_circuit(F, log2_n: int, rng_seed: int = 55):
    """instances of the five always-present library gates until the domain is full; rows a gate only READS as `next` are Zero rows"""
    import random
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gates import gate_rows, tables        # the satisfied instances the gate tests use (oracle/gates.py witness generators)
    p = F.p
    rnd = random.Random(rng_seed)
    gates, rows = [], []
    for r in range(4):                                # a few generic rows first: 1 * w0 - 5 = 0
        gates.append(CC.generic_gadget(p, r, CC.generic_spec(p, "Const", cst=5))); rows.append([5] + [0] * 14)
    full = False
    while not full:
        for name in ("Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar"):
            w, co, ngate = tables(name, rnd)
            if len(gates) + len(w) + 3 + 1 > (1 << log2_n) - 1:
                full = True
                break
            live = set(gate_rows(name, ngate))
            base = len(gates)
            for k, (wr, cr) in enumerate(zip(w, co)):
                typ = name if k in live else "Zero"
                gates.append(CC.gate(typ, base + k, [c % p for c in cr] if typ != "Zero" else []))
                rows.append(list(wr))
    cs = CC.build(F, gates)
    assert cs["log2_n"] == log2_n, cs["log2_n"]
    witness = [[r[c] for r in rows] for c in range(15)]
    CC.verify_witness(cs, witness)
    return cs, witness


def previous_challenges(C, srs, log_srs: int, count: int):
    """RecursionChallenge values as recursion.rs:56-70 makes them: random challenges, comm = commit_non_hiding(b_poly_coefficients(chals))"""
    std = P.StdRng(PREV_SEED)
    prev = []
    for _ in range(count):
        chals = [P.field_rand(C.scalar, std) for _ in range(log_srs)]
        prev.append((chals, srs.commit_non_hiding(P.b_poly_coefficients(C.scalar, chals), 1)))
    return prev


def make(name: str, verify: bool = True):
    cid, log2_n, log_srs, seed = CASES[name]
    C = P.CURVES[cid]; F = C.scalar
    t0 = time.time()
    nprev = 1 if name.endswith("_prev1") else 0
    srs = OPR.Srs(C, 1 << log_srs, threads=OPR.THREADS)
    if name.startswith("and_lookup"):
        cs, wrows = and_circuit(F, log2_n)
        rows, witness, desc = len(wrows), [[r[c] for r in wrows] for c in range(15)], "AND gadgets (create_and_witness)"
    elif name.startswith("library_gates"):
        cs, witness = library_circuit(F, log2_n)
        rows, desc = len(witness[0]), "instances of Poseidon, CompleteAdd, VarBaseMul, EndoMul, EndoMulScalar (oracle/gates.py witness generators)"
    elif name.startswith("generic_public"):
        cs, witness = generic_circuit(F, log2_n, log_srs)
        CC.verify_witness(cs, witness)
        rows, desc = len(witness[0]), "random 62-bit operands of addition / multiplication gates, 3 public inputs, copy constraints"
    else:
        cs, rows = bench_circuit(F, log2_n, log_srs)
        if nprev:
            cs["prev_challenges"] = nprev
        witness, desc = [[1] * rows for _ in range(15)], "15 columns of ones"
    prev = previous_challenges(C, srs, log_srs, nprev)
    ix = OPR.Index(C, cs, srs)
    t1 = time.time()
    proof = OPR.create_proof(ix, witness, P.StdRng(seed), prev_challenges=prev)
    t2 = time.time()
    raw = OPR.serialize_proof(C, proof)
    ok = None
    if verify:
        vix = dict(ix.vindex)
        if cs["public"]:                                   # the verifier's own rule (verifier.rs:834-858): from the inputs and the Lagrange-basis points
            xy, inf = srs.lagrange_basis(log2_n)[0]
            lag = [V.aff(C, xy[i], inf[i]) for i in range(cs["public"])]
            vix["public_comm"] = K.public_commitment(C, srs.h, lag, [witness[0][i] for i in range(cs["public"])])
        ok = bool(K.verify(C, vix, proof, None, srs.h, P.StdRng(bytes([5] * 32)), final_msm=V.final_msm_c(C, srs.g, srs.size)))
        assert ok, "the oracle verifier rejects the oracle prover's proof"
    rec = {"_generated_by": "tests/golden/make_proof_fixtures.py (oracle/prover.py; circuit kimchi/src/bench.rs:59-122, rng StdRng::from_seed(seed))",
           "name": name, "curve": ["vesta", "pallas"][cid], "log2_n": log2_n, "log2_srs": log_srs, "num_chunks": ix.num_chunks, "zk_rows": cs["zk_rows"],
           "gates": rows, "witness": desc, "prev_challenges": nprev, "lookup": cs["lookup"] is not None, "public": cs["public"], "seed_hex": seed.hex(), "proof_len": len(raw), "proof_sha256": hashlib.sha256(raw).hexdigest(),
           "verifier_index_digest_hex": hex(ix.digest), "accepted_by_oracle_verifier": ok,
           "challenges_hex": {k: (None if v is None else hex(v)) for k, v in proof["challenges"].items()},
           "seconds": {"index": round(t1 - t0, 1), "prove": round(t2 - t1, 1)}}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name + ".proof.bin"), "wb") as f:
        f.write(raw)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(name, rec["proof_len"], "bytes", rec["proof_sha256"][:16], rec["seconds"], flush=True)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or list(CASES)):
        make(nm)
