"""Row a12 as PARITY, not acceptance: the device prover (proof_systems_amd/prover.py over the C ABI) against bytes.

  * the reference's seeded whole-proof regression (kimchi/src/tests/and.rs:126-160, 404-731; golden copy
    tests/golden/and_serialization_regression.json): the device prover, drawing from the same StdRng stream as the Rust test,
    produces the 6160 serialised bytes the reference asserts -- Xor16 rows + their lookups + generic rows, a 2^9 domain under the
    2^16 SRS (16 folding rounds), every commitment, evaluation and the opening;
  * at other sizes / circuits / curves the device proof equals, byte for byte, the proof of the oracle's CPU prover
    (oracle/prover.py, itself pinned on the vector above) run on the same circuit, witness and random stream: chunked proofs
    (domain larger than the SRS, kimchi/src/tests/chunked.rs), previous challenges (recursion), public inputs, Pallas.
"""
import json
import os

import numpy as np
import pytest

from oracle import circuit as CC
from oracle import gates as G
from oracle import kimchi as K
from oracle import pasta as P
from oracle import prover as OPR
from oracle import views as V

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def _limbs(F, vals):
    from oracle import cref
    return cref.ints_to_limbs([F.to_mont(v % F.p) for v in vals])


def device_index(khip, cs, curve_id, srs, runtime_cfg=None):
    """A proof_systems_amd.prover.ProverIndex for a constraint system of oracle/circuit.py::build (the circuit description is the
    caller's in the reference too: gates, wires, coefficients)."""
    from proof_systems_amd import prover, lookup as LK
    F = cs["F"]
    rows = max(r for r, g in enumerate(cs["gates"]) if g["typ"] != "Zero") + 1
    co = np.stack([_limbs(F, [cs["coefficients"][c][r] for c in range(15)]) for r in range(rows)])
    ix = prover.ProverIndex(curve_id, cs["log2_n"], co, srs=srs, gate_types=cs["gate_types"][:rows], public=cs["public"], zk_rows=cs["zk_rows"])
    ix.set_wiring([g["wires"] for g in cs["gates"][:rows]])
    if cs["lookup"] is not None:
        ix.attach_lookup(LK.LookupIndex(ix.fid, cs["gate_types"], [], cs["log2_n"], cs["zk_rows"], runtime_tables=runtime_cfg))
    return ix


def compare(curve, oproof, dproof):
    """first differing component of two proofs in the oracle's representation, or None"""
    for k in ("w_comm", "z_comm", "t_comm"):
        if oproof[k] != dproof[k]:
            return k
    if (oproof.get("lookup") or {}) != (dproof.get("lookup") or {}):
        return "lookup commitments"
    eo, ed = K.normalize_evals(oproof["evals"]), K.normalize_evals(dproof["evals"])
    for k in eo:
        if k in ed and eo[k] != ed[k]:
            return "evals." + k
    if oproof["ft_eval1"] != dproof["ft_eval1"]:
        return "ft_eval1"
    for k in ("lr", "delta", "z1", "z2", "sg"):
        if oproof["opening"][k] != dproof["opening"][k]:
            return "opening." + k
    return None


def test_device_prover_reproduces_the_reference_whole_proof_bytes(khip):
    from proof_systems_amd import prover
    with open(os.path.join(HERE, "golden", "and_serialization_regression.json")) as f:
        kat = json.load(f)
    seed, want = bytes(kat["seed"]), bytes.fromhex(kat["proof_hex"])
    C = P.VESTA; F = C.scalar
    std = P.StdRng(seed)
    gates = []
    CC.extend_and(F.p, gates, 8)
    in1 = CC.gen_field_with_bits(std, 64); in2 = CC.gen_field_with_bits(std, 64)
    rows = G.and_witness(F, in1, in2, 8)
    cs = CC.build(F, gates)
    srs = khip.Srs.create(khip.VESTA, 1 << 16)
    ix = device_index(khip, cs, khip.VESTA, srs)
    wit = np.stack([_limbs(F, [r[c] for r in rows]) for c in range(15)])
    proof = prover.create_proof(ix, wit, V.RefRng(std))
    c, vix, pr = V.device_views(ix, proof)
    got = OPR.serialize_proof(C, pr)
    # ... and through kh_prove (host loop in C++, lookups included), from a fresh copy of the same random stream
    std2 = P.StdRng(seed); CC.gen_field_with_bits(std2, 64); CC.gen_field_with_bits(std2, 64)
    nproof = prover.create_proof_native(ix, wit, V.RefRng(std2))
    assert OPR.serialize_proof(C, V.device_views(ix, nproof)[2]) == want, "kh_prove differs from the reference's bytes"
    if got != want:
        import msgpack
        first = next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)
        ref = msgpack.unpackb(want, raw=True)
        mine = msgpack.unpackb(got, raw=True)
        where = [k for k, (a, b) in zip(("commitments", "opening", "evals", "ft_eval1", "prev"), zip(mine, ref)) if a != b]
        pytest.fail(f"device proof differs from the reference's bytes: first at offset {first}, in {where}")
    assert len(got) == 6160


@pytest.mark.parametrize("log_srs", [8, 7])
def test_chunked_proof_with_lookups_equals_the_oracle_provers_proof(khip, log_srs):
    """The two features together, which no reference fixture and no other test combines: the AND circuit of the whole-proof vector (Xor16 rows + XOR-table
    lookups, a 2^9 domain) over an SRS of 2^8 / 2^7 -- 2 / 4 chunks, zk_rows 5 / 9: every lookup commitment a chunk list, the sorted / aggregation /
    table evaluations chunked, the combined table's blinder per chunk.  Python loop and kh_prove = the oracle prover, byte for byte; verifier accepts."""
    from proof_systems_amd import prover
    C = P.VESTA; F = C.scalar
    std = P.StdRng(bytes([61] * 32))
    gates = []
    CC.extend_and(F.p, gates, 8)
    in1 = CC.gen_field_with_bits(std, 64); in2 = CC.gen_field_with_bits(std, 64)
    rows = G.and_witness(F, in1, in2, 8)
    cs = CC.build(F, gates, max_poly_size=1 << log_srs)
    nch = (1 << cs["log2_n"]) >> log_srs
    assert cs["log2_n"] == 9 and cs["zk_rows"] == (16 * nch + 5) // 7 and cs["lookup"] is not None
    wit = [[r[c] for r in rows] for c in range(15)]
    seed = bytes([62, log_srs] + [1] * 30)
    osrs = OPR.Srs(C, 1 << log_srs)
    oix = OPR.Index(C, cs, osrs)
    oproof = OPR.create_proof(oix, wit, P.StdRng(seed))
    assert K.verify(C, dict(oix.vindex), oproof, None, osrs.h, P.StdRng(bytes([5] * 32)), final_msm=V.final_msm_c(C, osrs.g, osrs.size))
    srs = khip.Srs.create(khip.VESTA, 1 << log_srs)
    ix = device_index(khip, cs, khip.VESTA, srs)
    assert ix.num_chunks == nch
    w = np.stack([_limbs(F, col) for col in wit])
    dproof = prover.create_proof(ix, w, V.RefRng(P.StdRng(seed)))
    c, vix, pr = V.device_views(ix, dproof)
    assert compare(C, oproof, pr) is None, compare(C, oproof, pr)
    assert OPR.serialize_proof(C, pr) == OPR.serialize_proof(C, oproof)
    nproof = prover.create_proof_native(ix, w, V.RefRng(P.StdRng(seed)))
    assert OPR.serialize_proof(C, V.device_views(ix, nproof)[2]) == OPR.serialize_proof(C, oproof)
    ix.free()


@pytest.mark.parametrize("cid,logn,log_srs", [(0, 7, 7), (1, 7, 7), (0, 10, 10), (0, 8, 10), (0, 9, 7), (1, 8, 7), (0, 12, 11)])
def test_device_proof_equals_the_oracle_provers_proof(khip, cid, logn, log_srs):
    """generic-gate circuits with copy constraints and public inputs; log_srs > logn: SRS longer than the domain; log_srs < logn:
    chunked (2^(logn - log_srs) chunks, more zero-knowledge rows)."""
    from proof_systems_amd import prover
    C = P.CURVES[cid]; F = C.scalar; p = F.p
    rnd = np.random.default_rng(100 * cid + logn)
    n = 1 << logn
    nch = 1 << max(0, logn - log_srs)
    zk = (16 * nch + 5) // 7
    rows = n - zk - 5
    npub = 3
    gates, wit = [], [[0] * rows for _ in range(15)]
    for r in range(rows):
        a, b = int(rnd.integers(1, 1 << 62)), int(rnd.integers(1, 1 << 62))
        if r < npub:
            gates.append(CC.generic_gadget(p, r, CC.generic_spec(p, "Pub")))
            wit[0][r] = a
        else:                                                   # a + b - sum = 0 and a' * b' - prod = 0
            gates.append(CC.generic_gadget(p, r, CC.generic_spec(p, "Add"), CC.generic_spec(p, "Mul")))
            wit[0][r], wit[1][r], wit[2][r] = a, b, (a + b) % p
            wit[3][r], wit[4][r], wit[5][r] = b, a, a * b % p
    for r in range(npub, rows - 1, 2):                          # copy constraints: (r, 0) ~ (r, 4) hold the same value a
        CC.connect_cell_pair(gates, (r, 0), (r, 4))
    cs = CC.build(F, gates, public=npub, max_poly_size=1 << log_srs)
    assert cs["log2_n"] == logn and cs["zk_rows"] == zk
    CC.verify_witness(cs, wit)
    seed = bytes([7 + logn, cid] + [3] * 30)
    osrs = OPR.Srs(C, 1 << log_srs)
    oix = OPR.Index(C, cs, osrs)
    oproof = OPR.create_proof(oix, wit, P.StdRng(seed))
    srs = khip.Srs.create(cid, 1 << log_srs)
    ix = device_index(khip, cs, cid, srs)
    c, vix, _ = V.device_views(ix, None)
    for k in ("sigma_comm", "coefficients_comm", "generic_comm", "psm_comm"):
        assert vix[k] == oix.vindex[k], k
    dproof = prover.create_proof(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)))
    c, vix, pr = V.device_views(ix, dproof)
    assert compare(C, oproof, pr) is None, compare(C, oproof, pr)
    assert OPR.serialize_proof(C, pr) == OPR.serialize_proof(C, oproof)
    # and the oracle's verifier accepts it (public commitment recomputed from the inputs by the verifier's rule)
    assert K.verify(C, vix, pr, None, vix["h"], P.StdRng(bytes([5] * 32)), final_msm=V.final_msm_c(C, ix.srs.get_g(), ix.size))
    ix.free()


@pytest.mark.parametrize("cid,logn,log_srs,nprev", [(0, 6, 7, 1), (1, 7, 7, 2), (0, 7, 6, 1)])
def test_recursive_proof_equals_the_oracle_provers_proof(khip, cid, logn, log_srs, nprev):
    """create_recursive with previous challenges (kimchi/src/tests/recursion.rs:44-75: chals random, comm = commit_non_hiding of
    b_poly_coefficients(chals)): absorbed into both sponges, their polynomials opened first."""
    from proof_systems_amd import prover
    C = P.CURVES[cid]; F = C.scalar; p = F.p
    # polynomials/generic.rs:380-470 (create_circuit(0, 0) + fill_in_witness) over this curve's scalar field
    gates = [CC.generic_gadget(p, r, CC.generic_spec(p, "Add", right=3), CC.generic_spec(p, "Mul", mul=2)) for r in range(10)]
    gates += [CC.generic_gadget(p, r, CC.generic_spec(p, "Const", cst=3), CC.generic_spec(p, "Const", cst=5)) for r in range(10, 20)]
    wit = [[0] * 20 for _ in range(15)]
    for r in range(10):
        wit[0][r], wit[1][r], wit[2][r] = 11, 23, 11 + 23 * 3
        wit[3][r], wit[4][r], wit[5][r] = 11, 23, 11 * 23 * 2
    for r in range(10, 20):
        wit[0][r], wit[3][r] = 3, 5
    cs = CC.build(F, gates + [CC.gate("Zero", len(gates) + k) for k in range((1 << logn) - 9 - len(gates))], prev_challenges=nprev,
                  max_poly_size=(1 << log_srs) if log_srs < logn else None)
    assert cs["log2_n"] == logn
    CC.verify_witness(cs, wit)
    osrs = OPR.Srs(C, 1 << log_srs)
    std = P.StdRng(bytes([11] * 32))
    prev = []
    for _ in range(nprev):
        # (logn > log_srs: the accumulator's polynomial is twice the SRS size -- its commitment has two chunks, its evaluations are chunked: proof.rs:455-494)
        chals = [P.field_rand(F, std) for _ in range(max(log_srs, logn))]
        prev.append((chals, osrs.commit_non_hiding(P.b_poly_coefficients(F, chals), 1)))
    seed = bytes([21, cid] + [5] * 30)
    oix = OPR.Index(C, cs, osrs)
    oproof = OPR.create_proof(oix, wit, P.StdRng(seed), prev_challenges=prev)
    assert K.verify(C, dict(oix.vindex), oproof, None, osrs.h, P.StdRng(bytes([5] * 32)), final_msm=V.final_msm_c(C, osrs.g, osrs.size))
    srs = khip.Srs.create(cid, 1 << log_srs)
    ix = device_index(khip, cs, cid, srs)
    B = C.base

    def dev_comm(chunks):
        xy = np.zeros((len(chunks), 8), dtype=np.uint64); inf = np.zeros(len(chunks), dtype=np.uint8)
        for j, q in enumerate(chunks):
            if q is None:
                inf[j] = 1
            else:
                xy[j] = _limbs(B, [q[0], q[1]]).reshape(8)
        return xy, inf
    dprev = [(chals, dev_comm(cm)) for chals, cm in prev]
    # the device computes the same accumulator commitment (kh_b_poly_coefficients + commit_non_hiding)
    bc = khip.b_poly_coefficients(ix.fid, _limbs(F, prev[0][0]), len(prev[0][0]))[0]
    com, inf = srs.commit_non_hiding(bc, 1)
    assert V.chunks(C, (com, inf)) == prev[0][1]
    dproof = prover.create_proof(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)), prev_challenges=dprev)
    c, vix, pr = V.device_views(ix, dproof)
    assert compare(C, oproof, pr) is None, compare(C, oproof, pr)
    assert OPR.serialize_proof(C, pr) == OPR.serialize_proof(C, oproof)
    # the same through kh_prove_recursive (host loop in C++)
    nproof = prover.create_proof_native(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)), prev_challenges=dprev)
    assert OPR.serialize_proof(C, V.device_views(ix, nproof)[2]) == OPR.serialize_proof(C, oproof)
    ix.free()


def test_optional_gate_circuit_proof_equals_the_oracle_provers_proof(khip):
    """A circuit of RangeCheck0 rows (12-bit lookups into the range-check table, the compact form on one of them), a Rot64 row with its
    two range-check rows, and a ForeignFieldAdd row: optional-gate selectors committed non-hiding and evaluated, the RangeCheck lookup
    pattern with its 2^12-entry table on a 2^13 domain -- device proof = oracle proof, byte for byte."""
    from proof_systems_amd import prover
    import random
    from test_gates import tables
    C = P.VESTA; F = C.scalar; p = F.p
    rnd = random.Random(77)
    gates, rows = [], []

    def put(name, coeff_rows, wit_rows, ngate):
        base = len(gates)
        for k, wr in enumerate(wit_rows):
            typ = name if k < ngate else "Zero"
            gates.append(CC.gate(typ, base + k, [c % p for c in coeff_rows[k]] if typ != "Zero" else []))
            rows.append(list(wr))
    for name in ("RangeCheck0", "Rot64", "ForeignFieldAdd", "RangeCheck0"):
        w, co, ngate = tables(name, rnd)
        if name == "Rot64":                                     # the rot row reads `shifted` in the next row, which is itself a RangeCheck0 row: decompose it
            sh = w[1][0]
            w[1] = [sh] + [(sh >> (76 - 12 * k)) & 4095 for k in range(6)] + [(sh >> (14 - 2 * k)) & 3 for k in range(8)]
            put("Rot64", co, w[:1], 1); put("RangeCheck0", [[0] * 15], [w[1]], 1)
        else:
            put(name, co, w, ngate)
    cs = CC.build(F, gates)
    assert cs["log2_n"] == 13 and cs["optional"] == ["RangeCheck0", "ForeignFieldAdd", "Rot64"] and cs["lookup"].info.patterns == ["RangeCheck"]
    wit = [[r[c] for r in rows] for c in range(15)]
    CC.verify_witness(cs, wit)
    seed = bytes([44] * 32)
    osrs = OPR.Srs(C, 1 << 13)
    oix = OPR.Index(C, cs, osrs)
    oproof = OPR.create_proof(oix, wit, P.StdRng(seed))
    assert K.verify(C, dict(oix.vindex), oproof, None, osrs.h, P.StdRng(bytes([5] * 32)), final_msm=V.final_msm_c(C, osrs.g, osrs.size))
    srs = khip.Srs.create(khip.VESTA, 1 << 13)
    ix = device_index(khip, cs, khip.VESTA, srs)
    c, vix, _ = V.device_views(ix, None)
    assert vix["optional_comms"] == oix.vindex["optional_comms"] and K.verifier_index_digest(C, vix | {"lookup_index": None}) is not None
    dproof = prover.create_proof(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)))
    c, vix, pr = V.device_views(ix, dproof)
    assert compare(C, oproof, pr) is None, compare(C, oproof, pr)
    assert OPR.serialize_proof(C, pr) == OPR.serialize_proof(C, oproof)
    nproof = prover.create_proof_native(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)))      # kh_prove: optional gates + RangeCheck lookups
    assert OPR.serialize_proof(C, V.device_views(ix, nproof)[2]) == OPR.serialize_proof(C, oproof)
    ix.free()


def runtime_table_circuit(F, seed=5):
    """kimchi/src/tests/lookup.rs:249-330 (test_runtime_table): 20 Lookup gates into five runtime tables with ids 1..5, first column
    [8, 9, 8, 7, 1] fixed in the index, second column [0, 2, 3, 4, 5] supplied with the proof."""
    import random
    rnd = random.Random(seed)
    first, data = [8, 9, 8, 7, 1], [0, 2, 3, 4, 5]
    cfg = [{"id": tid, "first_column": list(first)} for tid in range(1, 6)]
    rts = [(c["id"], list(data)) for c in cfg]
    gates = [CC.gate("Lookup", r) for r in range(20)]
    wit = [[0] * 20 for _ in range(15)]
    for r in range(20):
        wit[0][r] = rnd.randrange(1, 6)
        for k in range(3):
            idx = rnd.randrange(5)
            wit[1 + 2 * k][r], wit[2 + 2 * k][r] = first[idx], data[idx]
    return CC.build(F, gates, runtime_tables=cfg), cfg, rts, wit


@pytest.mark.parametrize("chunks", [1, 2])
def test_runtime_table_proof_equals_the_oracle_provers_proof(khip, chunks):
    """Runtime tables (prover.rs:397-470): the proof's contribution to the table's second column is committed first, enters the combined
    table through the joint combiner (blinders included), is evaluated and opened with its selector, and adds one lookup constraint.
    chunks = 2: the same over an SRS half the domain (the runtime column's commitment, blinders and evaluations per chunk)."""
    from proof_systems_amd import prover
    C = P.VESTA; F = C.scalar
    cs, cfg, rts, wit = runtime_table_circuit(F)
    if chunks > 1:
        from oracle import circuit as CCm
        cs = CCm.build(F, [CCm.gate("Lookup", r) for r in range(20)], runtime_tables=cfg, max_poly_size=(1 << cs["log2_n"]) // chunks)
        assert ((1 << cs["log2_n"]) // chunks) * chunks == 1 << cs["log2_n"]
    size = (1 << cs["log2_n"]) // chunks
    seed = bytes([61] * 32)
    osrs = OPR.Srs(C, size)
    oix = OPR.Index(C, cs, osrs)
    oproof = OPR.create_proof(oix, wit, P.StdRng(seed), runtime_tables=rts)
    assert K.verify(C, dict(oix.vindex), oproof, None, osrs.h, P.StdRng(bytes([5] * 32)), final_msm=V.final_msm_c(C, osrs.g, osrs.size))
    srs = khip.Srs.create(khip.VESTA, size)
    ix = device_index(khip, cs, khip.VESTA, srs, runtime_cfg=cfg)
    assert ix.num_chunks == chunks
    c, vix, _ = V.device_views(ix, None)
    assert K.verifier_index_digest(C, vix | {"lookup_index": None}) is not None
    dproof = prover.create_proof(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)), runtime_tables=rts)
    c, vix, pr = V.device_views(ix, dproof)
    assert vix["lookup_index"]["runtime_tables_selector"] == oix.vindex["lookup_index"]["runtime_tables_selector"]
    assert compare(C, oproof, pr) is None, compare(C, oproof, pr)
    assert OPR.serialize_proof(C, pr) == OPR.serialize_proof(C, oproof)
    with pytest.raises(ValueError):
        prover.create_proof(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)), runtime_tables=rts[:4])      # RuntimeTablesInconsistent
    # the same through kh_prove_full (host loop in C++): runtime column committed hiding, folded into the combined table, opened with its selector
    nproof = prover.create_proof_native(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)), runtime_tables=rts)
    assert OPR.serialize_proof(C, V.device_views(ix, nproof)[2]) == OPR.serialize_proof(C, oproof)
    with pytest.raises(ValueError):
        prover.create_proof_native(ix, np.stack([_limbs(F, col) for col in wit]), V.RefRng(P.StdRng(seed)), runtime_tables=rts[:4])
    nx = prover.native_index(ix)
    with pytest.raises(khip.KhError, match="RuntimeTablesInconsistent"):
        nx.prove(witness=np.stack([_limbs(F, col) for col in wit]), runtime=np.zeros((3, 4), np.uint64))
    ix.free()
