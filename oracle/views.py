"""TEST INFRASTRUCTURE ONLY: the device prover's index / proof (limb arrays straight off the C ABI, proof_systems_amd/prover.py)
as the plain-integer structures the oracle works with (oracle/kimchi.py::verify, oracle/prover.py::serialize_proof), and a
`field_elements` adapter that lets the device prover draw from the oracle's StdRng restatement (the Rust caller's `rng`)."""
import numpy as np

from . import cref
from . import kimchi as K
from . import pasta as P


def aff(curve: P.Curve, xy, inf):
    if inf:
        return None
    v = P.from_limbs(np.asarray(xy)[:4]), P.from_limbs(np.asarray(xy)[4:])
    return (curve.base.from_mont(v[0]), curve.base.from_mont(v[1]))


def chunks(curve: P.Curve, t):
    """(xy (k, 8) or (8,), inf (k,) or a flag) -> list of affine points / None"""
    xy = np.asarray(t[0], dtype=np.uint64).reshape(-1, 8)
    inf = np.asarray(t[1]).reshape(-1)
    return [aff(curve, xy[j], inf[j]) for j in range(xy.shape[0])]


def device_views(ix, proof):
    """(curve, vix, proof) for oracle.kimchi.verify from a proof_systems_amd.prover.ProverIndex and the dict create_proof returns."""
    c = P.CURVES[ix.curve]
    ch = lambda t: chunks(c, t)
    vix = {"F": c.scalar, "n": ix.n, "log2_n": ix.log2_n, "omega": ix.omega, "shifts": ix.shifts, "h": aff(c, ix.h, False),
           "max_poly_size": ix.size, "zk_rows": ix.zk_rows, "public": ix.public,
           "sigma_comm": [ch(t) for t in ix.sigma_comm], "coefficients_comm": [ch(t) for t in ix.coefficients_comm], "generic_comm": ch(ix.generic_comm),
           "psm_comm": ch(ix.selector_comms[0]), "complete_add_comm": ch(ix.selector_comms[1]), "mul_comm": ch(ix.selector_comms[2]),
           "emul_comm": ch(ix.selector_comms[3]), "endomul_scalar_comm": ch(ix.selector_comms[4]),
           "optional_comms": [ch(ix.optional_comms[t]) if t in ix.optional_comms else None for t in K.OPTIONAL_GATES]}
    if proof is None:
        return c, vix, None
    if ix.public:
        vix["public_comm"] = ch(proof["public_comm"])        # (a verifier recomputes it from the inputs: oracle.kimchi.public_commitment)
    op = proof["opening"]
    pr = {"w_comm": [ch(t) for t in proof["w_comm"]], "z_comm": ch(proof["z_comm"]), "t_comm": ch(proof["t_comm"]),
          "evals": proof["evals"], "ft_eval1": proof["ft_eval1"],
          "opening": {"lr": [(aff(c, xy[0], li[0]), aff(c, xy[1], li[1])) for xy, li in op["lr"]], "delta": aff(c, *op["delta"]), "z1": op["z1"], "z2": op["z2"],
                      "sg": aff(c, *op["sg"])},
          "prev_challenges": [(list(chals), ch(cm)) for chals, cm in proof.get("prev_challenges", [])], "lookup": None}
    LI = getattr(ix, "lookup", None)
    if LI is not None:
        vix["lookup_index"] = {"joint_lookup_used": LI.joint_lookup_used, "lookup_table": [ch(t) for t in LI.table_comm],
                               "lookup_selectors": {q: (ch(LI.selector_comm[q]) if q in LI.patterns else None) for q in K.LOOKUP_PATTERN_ORDER},
                               "table_ids": ch(LI.table_ids_comm) if LI.table_ids_comm else None, "max_per_row": LI.max_per_row,
                               "max_joint_size": LI.max_joint_size, "patterns": list(LI.patterns), "uses_runtime_tables": LI.runtime_selector is not None,
                               "runtime_tables_selector": ch(LI.runtime_selector_comm) if getattr(LI, "runtime_selector_comm", None) else None}
        rt = proof["lookup"].get("runtime")
        pr["lookup"] = {"sorted": [ch(t) for t in proof["lookup"]["sorted"]], "aggreg": ch(proof["lookup"]["aggreg"]), "runtime": ch(rt) if rt is not None else None}
    return c, vix, pr


def final_msm_c(curve: P.Curve, g_limbs, n_srs: int, threads: int = 16):
    """The verifier's one MSM (ipa.rs:452-502) in the C oracle: independent of the product."""
    F = curve.scalar
    cid = 0 if curve is P.VESTA else 1

    def final_msm(g_terms, pts, sc):
        gs = [0] * n_srs
        for w, chal in g_terms:
            for j, s in enumerate(P.b_poly_coefficients(F, chal)):
                gs[j] = (gs[j] + w * s) % F.p
        live = [(p, s) for p, s in zip(pts, sc) if p is not None]
        xy = np.concatenate([g_limbs[:n_srs], np.stack([cref.ints_to_limbs([curve.base.to_mont(p[0]), curve.base.to_mont(p[1])]).reshape(8) for p, _ in live])])
        scal = cref.ints_to_limbs([F.to_mont(s) for s in gs + [s for _, s in live]])
        _, inf = cref.msm(cid, xy, scal, threads=threads)
        return inf
    return final_msm


class RefRng:
    """What the device prover's `rng` argument needs (Fld.rand_many: field_elements(field_id, k)), over the oracle's restatement of
    rand 0.8.5 StdRng + ark-ff's Fp::rand -- the stream a Rust caller's `&mut StdRng` would produce."""

    def __init__(self, std: P.StdRng):
        self.std = std

    def field_elements(self, fid: int, k: int):
        F = P.Fp if fid == 0 else P.Fq
        return [P.field_rand(F, self.std) for _ in range(k)]
