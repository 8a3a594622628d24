"""TEST INFRASTRUCTURE ONLY -- reader for the reference's stored proof fixtures (kimchi/src/tests/fixtures/*.bin).

Each file is a msgpack (rmp_serde) `RawFixture` (kimchi/src/tests/fixtures.rs:16-28): proof bytes, verifier-index bytes (both
msgpack again), public inputs (ark compressed), feature flags, the endo scalar.  They were produced BY THE REFERENCE (feature
`save-test-proofs`, kimchi/src/tests/framework.rs:559-590) and are what its own prover-less test mode verifies
(tests/generic.rs:56-103), so they are golden vectors for everything a proof depends on: the Lagrange-basis commitments of the
index columns (interpolation + MSM), the coset shifts, the Fiat-Shamir transcript, ft_eval0, the opening.

Layouts (serde: structs as arrays in declaration order, `#[serde(skip)]` fields absent):
  VerifierIndex      kimchi/src/verifier_index.rs:59-158    [domain, max_poly_size, zk_rows, public, prev_challenges, sigma_comm[7],
                     coefficients_comm[15], generic, psm, complete_add, mul, emul, endomul_scalar, 6 optional gate commitments,
                     shift[7], lookup_index]
  ProverProof        kimchi/src/proof.rs:134-195            [commitments [w[15], z, t, lookup], opening [lr, delta, z1, z2, sg], evals, ft_eval1,
                     prev_challenges]
  ProofEvaluations   kimchi/src/proof.rs:51-115             [public, w[15], z, s[6], coefficients[15], 6 selectors, 6 optional selectors, lookup ...]
  PolyComm           [chunks]; points: ark compressed (33 bytes), field elements: 32 bytes little-endian.
  Radix2EvaluationDomain (ark-poly): size u64, log_size u32, size_as_field, size_inv, group_gen, group_gen_inv, offset, offset_inv, offset_pow_size.
"""
from typing import Any, Dict

import msgpack

from . import pasta as P


def _fe(b: bytes) -> int:
    assert len(b) == 32
    return int.from_bytes(b, "little")


def _comm(curve: P.Curve, c) -> list:
    (chunks,) = c                                            # PolyComm { chunks }
    return [curve.decompress(bytes(x)) for x in chunks]


def _opt(x, f):
    return None if x is None else f(x)


def load(path: str, curve: P.Curve) -> Dict[str, Any]:
    raw = msgpack.unpackb(open(path, "rb").read(), raw=True, strict_map_key=False)
    proof_b, vi_b, pub_b, npub, flags = raw[0], raw[1], raw[2], raw[3], raw[4]
    endo = raw[5] if len(raw) > 5 else None
    pr = msgpack.unpackb(bytes(proof_b), raw=True, strict_map_key=False)
    vi = msgpack.unpackb(bytes(vi_b), raw=True, strict_map_key=False)
    F = curve.scalar
    comm = lambda c: _comm(curve, c)
    # ---- verifier index
    d = bytes(vi[0])
    size = int.from_bytes(d[:8], "little"); log_size = int.from_bytes(d[8:12], "little")
    dom = [_fe(d[12 + 32 * i: 44 + 32 * i]) for i in range(7)]
    vindex = {
        "n": size, "log2_n": log_size, "size_inv": dom[1], "omega": dom[2], "omega_inv": dom[3],
        "max_poly_size": vi[1], "zk_rows": vi[2], "public": vi[3], "prev_challenges": vi[4],
        "sigma_comm": [comm(c) for c in vi[5]], "coefficients_comm": [comm(c) for c in vi[6]],
        "generic_comm": comm(vi[7]), "psm_comm": comm(vi[8]), "complete_add_comm": comm(vi[9]), "mul_comm": comm(vi[10]),
        "emul_comm": comm(vi[11]), "endomul_scalar_comm": comm(vi[12]),
        "optional_comms": [_opt(c, comm) for c in vi[13:19]],
        "shifts": [_fe(bytes(s)) for s in vi[19]], "lookup_index": None,
        "F": F,
    }
    if vi[20] is not None:                                   # LookupVerifierIndex (verifier_index.rs:35-56), LookupInfo / LookupFeatures / LookupPatterns (lookups.rs)
        li = vi[20]
        info = li[4]
        vindex["lookup_index"] = {
            "joint_lookup_used": bool(li[0]), "lookup_table": [comm(c) for c in li[1]],
            "lookup_selectors": dict(zip(("Xor", "Lookup", "RangeCheck", "ForeignFieldMul"), [_opt(c, comm) for c in li[2]])),
            "table_ids": _opt(li[3], comm), "max_per_row": info[0], "max_joint_size": info[1],
            "patterns": [q for q, on in zip(("Xor", "Lookup", "RangeCheck", "ForeignFieldMul"), info[2][0]) if on],
            "uses_runtime_tables": bool(info[2][2]), "runtime_tables_selector": _opt(li[5], comm),
        }
    # ---- proof
    cm, op, ev = pr[0], pr[1], pr[2]
    pe = lambda e: ([_fe(bytes(x)) for x in e[0]], [_fe(bytes(x)) for x in e[1]])      # PointEvaluations { zeta, zeta_omega }
    evals = {
        "public": _opt(ev[0], pe), "w": [pe(e) for e in ev[1]], "z": pe(ev[2]), "s": [pe(e) for e in ev[3]],
        "coefficients": [pe(e) for e in ev[4]],
        "generic_selector": pe(ev[5]), "poseidon_selector": pe(ev[6]), "complete_add_selector": pe(ev[7]), "mul_selector": pe(ev[8]),
        "emul_selector": pe(ev[9]), "endomul_scalar_selector": pe(ev[10]),
        "optional_gate_selectors": [_opt(e, pe) for e in ev[11:17]],           # range_check0/1, foreign_field_add/mul, xor, rot
        "lookup_aggregation": _opt(ev[17], pe), "lookup_table": _opt(ev[18], pe), "lookup_sorted": [_opt(e, pe) for e in ev[19]],
        "runtime_lookup_table": _opt(ev[20], pe), "runtime_lookup_table_selector": _opt(ev[21], pe),
        "lookup_selectors": dict(zip(("Xor", "Lookup", "RangeCheck", "ForeignFieldMul"), [_opt(e, pe) for e in ev[22:26]])),
    }
    proof = {
        "w_comm": [comm(c) for c in cm[0]], "z_comm": comm(cm[1]), "t_comm": comm(cm[2]),
        "lookup": _opt(cm[3], lambda l: {"sorted": [comm(c) for c in l[0]], "aggreg": comm(l[1]), "runtime": _opt(l[2], comm)}),
        "opening": {"lr": [(curve.decompress(bytes(l)), curve.decompress(bytes(r))) for l, r in op[0]], "delta": curve.decompress(bytes(op[1])),
                    "z1": _fe(bytes(op[2])), "z2": _fe(bytes(op[3])), "sg": curve.decompress(bytes(op[4]))},
        "evals": evals, "ft_eval1": _fe(bytes(pr[3])),
        "prev_challenges": [([_fe(bytes(x)) for x in rc[0]], comm(rc[1])) for rc in pr[4]],      # RecursionChallenge { chals, comm } (proof.rs:117-131)
    }
    # public inputs: ark compressed field elements, back to back
    pub = [int.from_bytes(bytes(pub_b)[32 * i: 32 * i + 32], "little") for i in range(npub)]
    return {"proof": proof, "vindex": vindex, "public": pub, "feature_flags": flags, "endo": _opt(endo, lambda e: _fe(bytes(e)))}


def oracle_views(fx, h):
    """(vix, proof) in the structures oracle/kimchi.py::verify takes: one chunk per polynomial, evaluations as (zeta, zeta omega)."""
    from . import kimchi as K
    vix = dict(fx["vindex"]); vix["h"] = h
    ev = fx["proof"]["evals"]
    one = lambda e: (e[0][0], e[1][0])
    pe = {k: one(ev[k]) for k in K.EVAL_ORDER}
    pe.update({"public": one(ev["public"]) if ev["public"] is not None else None, "w": [one(e) for e in ev["w"]], "s": [one(e) for e in ev["s"]],
               "coefficients": [one(e) for e in ev["coefficients"]]})
    for k in ("lookup_aggregation", "lookup_table", "runtime_lookup_table", "runtime_lookup_table_selector"):
        pe[k] = one(ev[k]) if ev[k] is not None else None
    pe["optional_gate_selectors"] = [one(e) if e is not None else None for e in ev["optional_gate_selectors"]]
    pe["lookup_sorted"] = [one(e) for e in ev["lookup_sorted"] if e is not None]
    pe["lookup_selectors"] = {q: one(e) for q, e in ev["lookup_selectors"].items() if e is not None}
    proof = dict(fx["proof"]); proof["evals"] = pe
    return vix, proof
