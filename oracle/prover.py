"""TEST INFRASTRUCTURE ONLY (see oracle/pasta.py): `ProverIndex::create` + `ProverProof::create_recursive` restated on the CPU
-- plain Python integers for the field work, the C oracle (oracle/pasta_ref.c) for the MSMs -- with the REFERENCE'S RNG DRAW
ORDER, so that a proof is a deterministic function of (circuit, witness, StdRng seed) exactly as in the reference, and
`serialize_proof` writes it the way rmp-serde does.

  ProverIndex / VerifierIndex    kimchi/src/prover_index.rs:60-160, verifier_index.rs:175-300 (commitments), :405-540 (digest)
  create_recursive               kimchi/src/prover.rs:187-1515 (draw order: zk rows :254-266 in REVERSE row order per column, witness
                                 blinders :316-327, sorted zk_patch + blinders :583-604, aggregation :640-660, perm_aggreg :679, z :682,
                                 t :923, opening ipa.rs:929-1040)
  ProverProof / ProofEvaluations kimchi/src/proof.rs:51-195 (serde layout)
  SRS::open                      oracle/pasta.py::ipa_open (pinned on the reference's opening-proof bytes); here with the C MSM as
                                 the engine of the L / R sums (the basis is kept unfolded, see `_Rounds`)

PINNED: `create_proof` on the circuit / inputs / seed of kimchi/src/tests/and.rs:126-160 reproduces the reference's
serialised proof (and.rs:412-727, 6160 bytes) byte for byte (tests/test_reference_kat.py); the index side equals the verifier
index the reference stored for the same circuit.  With that, this module is the oracle the device prover is compared with
value for value at other sizes and circuits (tests/test_gpu_prover_parity.py)."""
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import circuit as CC
from . import cref
from . import gates as G
from . import kimchi as K
from . import lookup as L
from . import pasta as P
from . import poseidon as S

COLUMNS, PERMUTS = 15, 7

# Transforms of 2^FAST_LOG points and more run in the C oracle (oracle/pasta_ref.c::ko_ntt, held to oracle/pasta.py::ntt point for point by
# tests/test_oracle_kats.py) instead of on Python integers: the same values, and what lets this prover reach BASELINE config 3's own size
# (2^16 gates: 8n = 2^19-point extensions) in minutes.  FAST_LOG = None puts everything back on Python integers; tests/test_reference_kat.py
# reproduces the reference's whole-proof bytes on BOTH paths.
FAST_LOG: Optional[int] = 8
THREADS = 16


def _fid(F) -> int:
    return 0 if F is P.Fp else 1


def to_limbs(F, vals: Sequence[int]) -> np.ndarray:
    """canonical integers -> (k, 4) Montgomery limbs (the C oracle's wire format)"""
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)
    return cref.field_op(_fid(F), "to_mont", a) if len(vals) else a.copy()


def from_limbs(F, a: np.ndarray) -> List[int]:
    b = cref.field_op(_fid(F), "from_mont", np.ascontiguousarray(a).reshape(-1, 4)).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def ntt(F, a: Sequence[int], log2_n: int, inverse: bool = False) -> List[int]:
    """oracle/pasta.py::ntt (ark-poly Radix2EvaluationDomain semantics), through the C oracle from 2^FAST_LOG points on."""
    if FAST_LOG is None or log2_n < FAST_LOG:
        return P.ntt(F, a, log2_n, inverse)
    n = 1 << log2_n
    assert len(a) <= n
    x = to_limbs(F, [v % F.p for v in a] + [0] * (n - len(a)))
    return from_limbs(F, cref.ntt(_fid(F), x, log2_n, inverse, threads=THREADS))


# ---------------------------------------------------------------------------------------------------- SRS + commitments
class Srs:
    """ipa::SRS: g (first `size` points of SRS::create), h, Lagrange bases per domain (computed by the C oracle's group iNTT)."""

    def __init__(self, curve: P.Curve, size: int, threads: int = 8):
        self.curve, self.cid, self.size, self.threads = curve, (0 if curve is P.VESTA else 1), size, threads
        self.g = cref.srs_generate(self.cid, 0, size, threads=threads)
        self.h = curve.srs_h()
        self.bases: Dict[int, list] = {}

    def _aff(self, xy, inf):
        if inf:
            return None
        v = cref.limbs_to_ints(np.asarray(xy).reshape(2, 4))
        B = self.curve.base
        return (B.from_mont(v[0]), B.from_mont(v[1]))

    def msm(self, points_xy, scalars: Sequence[int], inf=None):
        F = self.curve.scalar
        sc = to_limbs(F, [s % F.p for s in scalars])
        xy, i = cref.msm(self.cid, points_xy, sc, inf=inf, threads=self.threads)
        return self._aff(xy, i)

    def lagrange_basis(self, log2_n: int):
        """SRS::lagrange_basis (ipa.rs:1065-1172): per chunk, the group iNTT over that chunk's share of g."""
        if log2_n not in self.bases:
            n = 1 << log2_n
            chunks = max(1, n // self.size)
            self.bases[log2_n] = [cref.lagrange_basis(self.cid, self.g[:min(n, self.size)], log2_n, c) for c in range(chunks)]
        return self.bases[log2_n]

    def commit_evaluations_non_hiding(self, log2_n: int, evals: Sequence[int]) -> list:
        """ipa.rs:706-728: one MSM per chunk of the basis, each over ALL the evaluations."""
        return [self.msm(xy, evals, inf=inf) for xy, inf in self.lagrange_basis(log2_n)]

    def commit_non_hiding(self, coeffs: Sequence[int], num_chunks: int) -> list:
        """ipa.rs:638-683."""
        q = self.curve.scalar.p
        coeffs = [c % q for c in coeffs]
        while coeffs and coeffs[-1] == 0:
            coeffs.pop()
        if not coeffs:
            chunks = [None]
        else:
            chunks = [self.msm(self.g[:len(coeffs[i:i + self.size])], coeffs[i:i + self.size]) for i in range(0, len(coeffs), self.size)]
        return chunks + [None] * (num_chunks - len(chunks))

    def mask(self, com: list, blinders: Sequence[int]) -> list:
        return P.mask_custom(self.curve, self.h, com, blinders)


# ---------------------------------------------------------------------------------------------------- index
class Index:
    """ProverIndex over a constraint system of oracle/circuit.py::build: verifier-index commitments and the digest."""

    def __init__(self, curve: P.Curve, cs, srs: Srs):
        self.curve, self.cs, self.srs = curve, cs, srs
        F = curve.scalar
        assert cs["F"] is F
        n, logn = cs["n"], cs["log2_n"]
        self.max_poly_size = srs.size
        self.num_chunks = 1 if n < srs.size else n // srs.size
        ce = lambda col: srs.commit_evaluations_non_hiding(logn, col)
        one = [1] * self.num_chunks
        v = {"F": F, "n": n, "log2_n": logn, "omega": cs["omega"], "shifts": cs["shifts"], "h": srs.h, "max_poly_size": srs.size, "zk_rows": cs["zk_rows"],
             "public": cs["public"], "prev_challenges": cs["prev_challenges"]}
        v["sigma_comm"] = [ce(c) for c in cs["sigma"]]
        v["coefficients_comm"] = [ce(c) for c in cs["coefficients"]]
        for key, name in (("generic_comm", "Generic"), ("psm_comm", "Poseidon"), ("complete_add_comm", "CompleteAdd"), ("mul_comm", "VarBaseMul"),
                          ("emul_comm", "EndoMul"), ("endomul_scalar_comm", "EndoMulScalar")):
            v[key] = srs.mask(ce(cs["selectors"][name]), one)
        v["optional_comms"] = [ce(cs["selectors"][t]) if t in cs["optional"] else None for t in K.OPTIONAL_GATES]
        v["lookup_index"] = None
        lcs = cs["lookup"]
        if lcs is not None:                                  # verifier_index.rs:189-216
            v["lookup_index"] = {
                "joint_lookup_used": lcs.info.joint_lookup_used, "lookup_table": [srs.mask(ce(c), one) for c in lcs.table_cols],
                "lookup_selectors": {q: (ce(lcs.selectors[q]) if q in lcs.info.patterns else None) for q in K.LOOKUP_PATTERN_ORDER},
                "table_ids": srs.mask(ce(lcs.table_ids), one) if lcs.table_ids is not None else None,
                "max_per_row": lcs.info.max_per_row, "max_joint_size": lcs.info.max_joint_size, "patterns": list(lcs.info.patterns),
                "uses_runtime_tables": lcs.runtime_selector is not None,
                "runtime_tables_selector": ce(lcs.runtime_selector) if lcs.runtime_selector is not None else None}
        self.vindex = v
        self.digest = K.verifier_index_digest(curve, v)


# ---------------------------------------------------------------------------------------------------- helpers
def _horner(p: int, coeffs: Sequence[int], x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % p
    return acc


def evaluate_chunks(p: int, coeffs: Sequence[int], x: int, num_chunks: int, size: int) -> List[int]:
    """DensePolynomial::to_chunked_polynomial(num_chunks, size).evaluate_chunks(x) (utils/src/dense_polynomial.rs:50-69)."""
    coeffs = _trim(list(coeffs))
    out = [_horner(p, coeffs[i:i + size], x) for i in range(0, len(coeffs), size)]
    assert len(out) <= num_chunks
    return out + [0] * (num_chunks - len(out))


def linearize_chunks(p: int, coeffs: Sequence[int], num_chunks: int, size: int, zeta_n: int) -> List[int]:
    """to_chunked_polynomial(num_chunks, size).linearize(zeta_n) (chunked_polynomial.rs:33-51)."""
    out = [0] * size
    scale = 1
    coeffs = _trim(list(coeffs))
    assert len(coeffs) <= num_chunks * size
    for i in range(0, len(coeffs), size):
        for k, c in enumerate(coeffs[i:i + size]):
            out[k] = (out[k] + scale * c) % p
        scale = scale * zeta_n % p
    return _trim(out)


def _trim(c: List[int]) -> List[int]:
    while c and c[-1] == 0:
        c.pop()
    return c


def _divide_by_linear(p: int, f: Sequence[int], a: int):
    """(quotient, remainder) of f / (x - a)."""
    q = [0] * max(len(f) - 1, 0)
    acc = 0
    for i in reversed(range(len(f))):
        acc = (acc * a + f[i]) % p
        if i:
            q[i - 1] = acc
    return q, acc


class _Rounds:
    """The L / R sums of SRS::open's folding loop (ipa.rs:929-1018) with the C oracle's MSM as the engine.  The basis is never folded:
    after j rounds g_j[i] = sum_k coef_j[k] * G[i + k * N_j] with coef_(j+1)[2k + b] = coef_j[k] * u^b, so every L, R and the final
    g0 is one MSM over the original SRS -- the same group elements as the literal loop of oracle/pasta.py::ipa_open_rounds (which
    is pinned on the reference's opening-proof bytes at n = 128), computed in seconds at n = 2^16."""

    def __init__(self, srs: Srs, a, b, u_base):
        self.srs, self.F = srs, srs.curve.scalar
        self.a, self.b, self.coef = list(a), list(b), [1]
        B = srs.curve.base
        extra = cref.ints_to_limbs([B.to_mont(srs.h[0]), B.to_mont(srs.h[1]), B.to_mont(u_base[0]), B.to_mont(u_base[1])]).reshape(2, 8)
        self.pts = np.concatenate([srs.g, extra])

    def round_lr(self, rand_l, rand_r):
        p = self.F.p
        a, b, coef = self.a, self.b, self.coef
        nj = len(a); m = nj // 2
        ip_l = sum(x * y for x, y in zip(a[m:], b[:m])) % p
        ip_r = sum(x * y for x, y in zip(a[:m], b[m:])) % p
        sl, sr = [], []
        for c in coef:
            sl += [x * c % p for x in a[m:]] + [0] * m
            sr += [0] * m + [x * c % p for x in a[:m]]
        return self.srs.msm(self.pts, sl + [rand_l, ip_l]), self.srs.msm(self.pts, sr + [rand_r, ip_r])

    def round_fold(self, u_pre):
        F = self.F; p = F.p
        _, endo_r = P.endos(self.srs.curve)
        u = P.challenge_to_field(F, u_pre, endo_r); ui = F.inv(u)
        m = len(self.a) // 2
        self.a = [(lo + ui * hi) % p for lo, hi in zip(self.a[:m], self.a[m:])]
        self.b = [(lo + u * hi) % p for lo, hi in zip(self.b[:m], self.b[m:])]
        self.coef = [x for c in self.coef for x in (c, c * u % p)]
        return u

    def finish(self):
        return self.a[0], self.b[0], self.srs.msm(self.srs.g, self.coef)


def unnormalized_lagrange_poly(p: int, omega: int, n: int, i: int) -> List[int]:
    """(x^n - 1) / (x - w^i) in coefficient form: sum_k w^(i (n - 1 - k)) x^k."""
    wi = pow(omega, i % n, p)
    out = [0] * n
    acc = 1
    for k in reversed(range(n)):
        out[k] = acc
        acc = acc * wi % p
    return out


# ---------------------------------------------------------------------------------------------------- the prover
def create_proof(ix: Index, witness: Sequence[Sequence[int]], rng: P.StdRng, prev_challenges=(), trace: Optional[dict] = None, runtime_tables=()):
    """ProverProof::create_recursive (prover.rs:187-1515).  witness: 15 columns of equal length <= n - zk_rows;
    prev_challenges: [(chals, comm chunks)]; runtime_tables: [(id, data)] in the configured order (RuntimeTable, runtime_tables.rs:52-58).  Returns the proof as plain integers / affine points (None = infinity):
    w_comm, z_comm, t_comm, lookup, opening, evals (each a pair of chunk lists), ft_eval1, prev_challenges."""
    curve, cs, srs = ix.curve, ix.cs, ix.srs
    F = curve.scalar; p = F.p
    n, logn, zk = cs["n"], cs["log2_n"], cs["zk_rows"]
    size, nch = srs.size, ix.num_chunks
    omega = cs["omega"]
    _, endo_r = P.endos(curve)
    rand = lambda: P.field_rand(F, rng)
    tr = trace if trace is not None else {}
    assert zk > CC.zk_rows_strict_lower_bound(nch), "NotZeroKnowledge"
    rows = len(witness[0])
    assert all(len(c) == rows for c in witness), "WitnessCsInconsistent"
    assert n - rows >= zk, "NoRoomForZkInWitness"
    # ---- pad; the last zk rows of every column are random, drawn from the LAST row backwards (prover.rs:254-266)
    w = []
    for col in witness:
        c = [v % p for v in col] + [0] * (n - rows)
        for r in range(n - 1, n - 1 - zk, -1):
            c[r] = rand()
        w.append(c)
    fq = S.DefaultFqSponge(curve)
    fq.absorb_fq([ix.digest])
    for chals, comm in prev_challenges:
        fq.absorb_g(comm)
    # ---- the negated public-input polynomial, committed non-hiding and masked with ones (prover.rs:281-309)
    pub = w[0][:cs["public"]]
    public_poly = _trim(ntt(F, [(-x) % p for x in pub] + [0] * (n - len(pub)), logn, inverse=True))
    public_comm = srs.mask(srs.commit_non_hiding(public_poly, nch), [1] * nch)
    fq.absorb_g(public_comm)
    # ---- witness commitments over the Lagrange basis, blinded (prover.rs:316-363)
    w_blind = [[rand() for _ in range(nch)] for _ in range(COLUMNS)]
    w_comm = [srs.mask(srs.commit_evaluations_non_hiding(logn, w[i]), w_blind[i]) for i in range(COLUMNS)]
    for c in w_comm:
        fq.absorb_g(c)
    tr["w_comm"] = w_comm
    w_poly = [ntt(F, w[i], logn, inverse=True) for i in range(COLUMNS)]
    # ---- lookup: joint combiner, combined table, sorted columns (prover.rs:383-633)
    lcs = cs["lookup"]
    lk = None
    types = cs["gate_types"]
    if lcs is not None:
        rt = None
        if lcs.runtime_selector is not None:                   # prover.rs:397-470
            assert [(i, len(d)) for i, d in runtime_tables] == lcs.runtime_tables, "RuntimeTablesInconsistent"
            rte = [0] * n
            off = lcs.runtime_offset
            for _id, data in runtime_tables:
                rte[off:off + len(data)] = [v % p for v in data]
                off += len(data)
            for r in range(n - 1, n - 1 - zk, -1):             # zero-knowledge rows, from the last row backwards
                rte[r] = rand()
            rt_poly = ntt(F, rte, logn, inverse=True)
            com = srs.commit_non_hiding(rt_poly, nch)
            rt_blind = [rand() for _ in com]
            rt_comm = srs.mask(com, rt_blind)
            fq.absorb_g(rt_comm)
            rt = {"evals": rte, "poly": rt_poly, "blind": rt_blind, "comm": rt_comm, "sel_poly": ntt(F, lcs.runtime_selector, logn, inverse=True)}
        jc = P.challenge_to_field(F, fq.challenge() if lcs.info.joint_lookup_used else 0, endo_r)
        table = lcs.joint_table(jc, rt["evals"] if rt else None)
        table_poly = ntt(F, table, logn, inverse=True)
        srt = L.sorted_columns(lcs, types, w, jc, table=table)
        srt = [L.zk_patch(c, n, zk, [rand() for _ in range(zk)]) for c in srt]
        s_blind, s_comm = [], []
        for c in srt:                                          # commit_evaluations(d1, v, rng): non-hiding, then one blinder per chunk
            com = srs.commit_evaluations_non_hiding(logn, c)
            bl = [rand() for _ in com]
            s_blind.append(bl); s_comm.append(srs.mask(com, bl))
        for c in s_comm:
            fq.absorb_g(c)
        lk = {"jc": jc, "table": table, "table_poly": table_poly, "sorted": srt, "s_blind": s_blind, "s_comm": s_comm, "rt": rt,
              "sorted_poly": [ntt(F, c, logn, inverse=True) for c in srt]}
    beta = fq.challenge(); gamma = fq.challenge()
    if lk is not None:                                         # prover.rs:635-673
        agg = L.aggregation(lcs, types, w, lk["jc"], beta, gamma, lk["sorted"], None, draw=rand, table=lk["table"])
        assert agg[n - zk - 1] == 1, "aggregation incorrect"
        com = srs.commit_evaluations_non_hiding(logn, agg)
        a_blind = [rand() for _ in com]
        a_comm = srs.mask(com, a_blind)
        fq.absorb_g(a_comm)
        lk.update({"agg": agg, "a_blind": a_blind, "a_comm": a_comm, "agg_poly": ntt(F, agg, logn, inverse=True)})
    # ---- permutation accumulator (permutation.rs:447-577): the two random rows are drawn inside the running product
    z = _perm_aggreg(F, w, cs["sigma"], cs["shifts"], cs["sid"], beta, gamma, zk, rand)
    assert z[n - zk] == 1, "Permutation: final value"
    z_poly = ntt(F, z, logn, inverse=True)
    com = srs.commit_non_hiding(z_poly, nch)
    z_blind = [rand() for _ in com]
    z_comm = srs.mask(com, z_blind)
    fq.absorb_g(z_comm)
    tr["z_comm"] = z_comm
    alpha = P.challenge_to_field(F, fq.challenge(), endo_r)
    # ---- quotient: every constraint on d8, interpolated, + public, / Z_H, + the boundary quotients (prover.rs:794-920)
    quotient = _quotient(ix, w_poly, z_poly, public_poly, lk, alpha, beta, gamma)
    com = srs.commit_non_hiding(quotient, 7 * nch)
    t_blind = [rand() for _ in com]
    t_comm = srs.mask(com, t_blind)
    assert len(t_comm) == 7 * nch
    fq.absorb_g(t_comm)
    tr["t_comm"] = t_comm
    zeta = P.challenge_to_field(F, fq.challenge(), endo_r)
    zetaw = zeta * omega % p
    # ---- chunked evaluations (prover.rs:942-1130)
    ec = lambda poly: (evaluate_chunks(p, poly, zeta, nch, size), evaluate_chunks(p, poly, zetaw, nch, size))
    interp = lambda col: ntt(F, col, logn, inverse=True)
    sel_poly = {k: interp(v) for k, v in cs["selectors"].items()}
    coef_poly = [interp(c) for c in cs["coefficients"]]
    sigma_poly = [interp(c) for c in cs["sigma"]]
    ev = {"public": ec(public_poly), "w": [ec(q) for q in w_poly], "z": ec(z_poly), "s": [ec(q) for q in sigma_poly[:PERMUTS - 1]],
          "coefficients": [ec(q) for q in coef_poly],
          "generic_selector": ec(sel_poly["Generic"]), "poseidon_selector": ec(sel_poly["Poseidon"]), "complete_add_selector": ec(sel_poly["CompleteAdd"]),
          "mul_selector": ec(sel_poly["VarBaseMul"]), "emul_selector": ec(sel_poly["EndoMul"]), "endomul_scalar_selector": ec(sel_poly["EndoMulScalar"]),
          "optional_gate_selectors": [ec(sel_poly[t]) if t in cs["optional"] else None for t in K.OPTIONAL_GATES],
          "lookup_aggregation": None, "lookup_table": None, "lookup_sorted": [], "lookup_selectors": {}}
    lk_sel_poly = {}
    if lk is not None:
        ev["lookup_aggregation"] = ec(lk["agg_poly"]); ev["lookup_table"] = ec(lk["table_poly"])
        ev["lookup_sorted"] = [ec(q) for q in lk["sorted_poly"]]
        lk_sel_poly = {q: interp(lcs.selectors[q]) for q in lcs.info.patterns}
        ev["lookup_selectors"] = {q: ec(lk_sel_poly[q]) for q in lcs.info.patterns}
        if lk["rt"] is not None:
            ev["runtime_lookup_table"] = ec(lk["rt"]["poly"]); ev["runtime_lookup_table_selector"] = ec(lk["rt"]["sel_poly"])
    zeta_srs = pow(zeta, size, p); zetaw_srs = pow(zetaw, size, p)
    zeta_n = pow(zeta, n, p)
    comb = lambda e: (_horner(p, e[0], zeta_srs), _horner(p, e[1], zetaw_srs))          # ProofEvaluations::combine (proof.rs:430-470)
    # ---- ft (prover.rs:1147-1200): the linearization has no un-evaluated column except sigma_6
    alphas = [pow(alpha, K.ALPHA_PERM0 + i, p) for i in range(3)]
    zkp = K.eval_permutation_vanishing_polynomial({"F": F, "n": n, "omega": omega, "zk_rows": zk}, zeta)
    scal = comb(ev["z"])[1] * beta % p * alphas[0] % p * zkp % p
    for we, se in zip(ev["w"], ev["s"]):
        scal = scal * ((gamma + beta * comb(se)[0] + comb(we)[0]) % p) % p
    scal = (-scal) % p
    f_lin = linearize_chunks(p, [scal * c % p for c in sigma_poly[PERMUTS - 1]], nch, size, zeta_srs)
    t_lin = linearize_chunks(p, quotient, 7 * nch, size, zeta_srs)
    m1 = (zeta_n - 1) % p
    ft = [((f_lin[i] if i < len(f_lin) else 0) - m1 * (t_lin[i] if i < len(t_lin) else 0)) % p for i in range(max(len(f_lin), len(t_lin)))]
    blinding_ft = (-m1 * _horner(p, t_blind, zeta_srs)) % p
    ft_eval1 = _horner(p, ft, zetaw)
    # ---- Fr-sponge (prover.rs:1206-1250, plonk_sponge.rs:60-160)
    fq_before = fq.clone()
    fr = S.ArithmeticSponge(F)
    dg = fq.clone().challenge_fq()
    fr.absorb([dg if dg < p else 0])
    prev = S.ArithmeticSponge(F)
    for chals, _ in prev_challenges:
        prev.absorb(list(chals))
    fr.absorb([prev.squeeze()])
    fr.absorb([ft_eval1])
    fr.absorb(ev["public"][0]); fr.absorb(ev["public"][1])
    order = [ev[k] for k in K.EVAL_ORDER] + ev["w"] + ev["coefficients"] + ev["s"] + [e for e in ev["optional_gate_selectors"] if e is not None]
    lk_sponge, lk_open = [], []
    if lk is not None:
        sels = [ev["lookup_selectors"][q] for q in K.LOOKUP_PATTERN_ORDER if q in ev["lookup_selectors"]]
        rts = [ev["runtime_lookup_table"], ev["runtime_lookup_table_selector"]] if lk["rt"] is not None else []
        lk_sponge = [ev["lookup_aggregation"], ev["lookup_table"]] + ev["lookup_sorted"] + rts + sels
    for e in order + lk_sponge:
        fr.absorb(e[0]); fr.absorb(e[1])
    v = P.challenge_to_field(F, fr.squeeze() & ((1 << 128) - 1), endo_r)
    u = P.challenge_to_field(F, fr.squeeze() & ((1 << 128) - 1), endo_r)
    # ---- the opening (prover.rs:1252-1490): polynomials with their blinders, in the reference's order
    ones, zeros = [1] * nch, [0] * nch
    plnms = [(P.b_poly_coefficients(F, chals), [0] * len(comm)) for chals, comm in prev_challenges]
    plnms += [(public_poly, ones), (ft, [blinding_ft]), (z_poly, z_blind)]
    plnms += [(sel_poly[k], ones) for k in ("Generic", "Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar")]
    plnms += [(w_poly[i], w_blind[i]) for i in range(COLUMNS)]
    plnms += [(q, zeros) for q in coef_poly] + [(q, zeros) for q in sigma_poly[:PERMUTS - 1]]
    plnms += [(sel_poly[t], zeros) for t in K.OPTIONAL_GATES if t in cs["optional"]]
    if lk is not None:
        plnms += [(q, bl) for q, bl in zip(lk["sorted_poly"], lk["s_blind"])] + [(lk["agg_poly"], lk["a_blind"])]
        jc_, tic_ = lcs.combiners(lk["jc"])
        base = 0 if not lcs.table_cols else 1
        for _ in range(1, len(lcs.table_cols)):
            base = (1 + jc_ * base) % p
        if lk["rt"] is not None:                               # prover.rs:1402-1432: the runtime column's blinders enter the table's through the joint combiner
            plnms.append((lk["table_poly"], [(jc_ * b_ + base + tic_) % p for b_ in lk["rt"]["blind"]]))
            plnms.append((lk["rt"]["poly"], lk["rt"]["blind"]))
            plnms.append((lk["rt"]["sel_poly"], zeros))
        else:
            plnms.append((lk["table_poly"], [(base + tic_) % p] * nch))
        plnms += [(lk_sel_poly[q], zeros) for q in K.LOOKUP_PATTERN_ORDER if q in lk_sel_poly]
    opening = P.ipa_open(curve, [None] * size, srs.h, plnms, [zeta, zetaw], v, u, fq_before, rng,
                         rounds_backend=lambda a, b, u_base: _Rounds(srs, a, b, u_base))
    proof = {"w_comm": w_comm, "z_comm": z_comm, "t_comm": t_comm, "evals": ev, "ft_eval1": ft_eval1,
             "opening": {k: opening[k] for k in ("lr", "delta", "z1", "z2", "sg")},
             "lookup": None if lk is None else {"sorted": lk["s_comm"], "aggreg": lk["a_comm"], "runtime": lk["rt"]["comm"] if lk["rt"] else None},
             "prev_challenges": [(list(c), list(m)) for c, m in prev_challenges],
             "challenges": {"beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v, "u": u, "joint_combiner": lk["jc"] if lk else None}}
    return proof


def _perm_aggreg(F, w, sigma, shifts, sid, beta, gamma, zk, rand):
    p = F.p
    n = len(sid)
    z = [1] * n
    for j in range(n - 1):
        if j != n - zk and j != n - zk + 1:
            num = den = 1
            for i in range(PERMUTS):
                num = num * ((w[i][j] + sid[j] * beta % p * shifts[i] + gamma) % p) % p
                den = den * ((w[i][j] + sigma[i][j] * beta + gamma) % p) % p
            z[j + 1] = z[j] * num % p * F.inv(den) % p
        else:
            z[j + 1] = rand()
    return z


def _quotient(ix: Index, w_poly, z_poly, public_poly, lk, alpha, beta, gamma) -> List[int]:
    curve, cs = ix.curve, ix.cs
    F = curve.scalar; p = F.p
    n, logn, zk = cs["n"], cs["log2_n"], cs["zk_rows"]
    omega = cs["omega"]
    n8 = 8 * n
    lde = lambda poly: ntt(F, list(poly), logn + 3)
    interp = lambda col: ntt(F, col, logn, inverse=True)
    w8 = [lde(q) for q in w_poly]
    z8 = lde(z_poly)
    co8 = [lde(interp(c)) for c in cs["coefficients"]]
    sg8 = [lde(interp(c)) for c in cs["sigma"]]
    live = [t for t in ("Generic", "Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar") + tuple(cs["optional"]) if any(cs["selectors"][t])]
    sel8 = {t: lde(interp(cs["selectors"][t])) for t in live}
    w8n = F.root_of_unity(logn + 3)
    x8 = [1] * n8
    for i in range(1, n8):
        x8[i] = x8[i - 1] * w8n % p
    alphas = [pow(alpha, K.ALPHA_PERM0 + i, p) for i in range(3)]
    zk_roots = [pow(omega, n - zk, p), pow(omega, n - zk + 1, p), pow(omega, n - 1, p)]
    mds = S.params("fp" if F is P.Fp else "fq")["mds"]
    endo = P.endos(P.PALLAS if curve is P.VESTA else P.VESTA)[0]
    lkc = None
    if lk is not None:
        lcs = cs["lookup"]
        lkc = {"sorted": [lde(q) for q in lk["sorted_poly"]], "aggreg": [lde(lk["agg_poly"])], "table": [lde(lk["table_poly"])],
               "sel": {q: lde(interp(lcs.selectors[q])) for q in lcs.info.patterns},
               "l0": lde(unnormalized_lagrange_poly(p, omega, n, 0)), "lfinal": lde(unnormalized_lagrange_poly(p, omega, n, -(zk + 1)))}
        lalpha = [pow(alpha, K.ALPHA_LOOKUP0 + k, p) for k in range(16)]
        if lk["rt"] is not None:
            lkc["rt"] = lde(lk["rt"]["poly"]); lkc["rtsel"] = lde(lk["rt"]["sel_poly"])
    f8 = [0] * n8
    for i in range(n8):
        x = x8[i]
        nx = (i + 8) % n8
        curr = [c[i] for c in w8]
        acc = 0
        # permutation (permutation.rs:237-283)
        a = z8[i]; b = z8[nx]
        for k in range(PERMUTS):
            a = a * ((curr[k] + gamma + x * beta % p * cs["shifts"][k]) % p) % p
            b = b * ((curr[k] + gamma + sg8[k][i] * beta) % p) % p
        zkpm = (x - zk_roots[0]) * (x - zk_roots[1]) % p * (x - zk_roots[2]) % p
        acc = (a - b) * alphas[0] % p * zkpm % p
        if live:
            nxt = [c[nx] for c in w8]
            co = [c[i] for c in co8]
            for t in live:
                s = sel8[t][i]
                if not s:
                    continue
                if t == "Generic":
                    c1 = (co[0] * curr[0] + co[1] * curr[1] + co[2] * curr[2] + co[3] * curr[0] * curr[1] + co[4]) % p
                    c2 = (co[5] * curr[3] + co[6] * curr[4] + co[7] * curr[5] + co[8] * curr[3] * curr[4] + co[9]) % p
                    acc = (acc + s * (c1 + alpha * c2)) % p
                else:
                    acc = (acc + s * G.combined_row(F, t, curr, nxt, co, alpha, mds=mds, endo=endo)) % p
        if lkc is not None:
            def cell(kind, idx, row, i=i, nx=nx):
                j = nx if row else i
                if kind == "w":
                    return w8[idx][j]
                if kind == "selector":
                    return lkc["sel"][idx][j]
                return lkc[kind][idx][j]
            vanish = 1
            for k in range(n - zk - 1, n):
                vanish = vanish * ((x - pow(omega, k, p)) % p) % p
            atoms = {"vanish": vanish, "l0": lkc["l0"][i], "lfinal": lkc["lfinal"][i]}
            vals = L.constraint_values(cs["lookup"], lk["jc"], beta, gamma, cell, atoms)
            if lk["rt"] is not None:                           # runtime_tables::constraints, at position 3 + 4 (constraints.rs:658-680)
                vals.append(lkc["rt"][i] * lkc["rtsel"][i] % p)
            for k, val in enumerate(vals):
                acc = (acc + lalpha[k] * val) % p
        f8[i] = acc
    f = ntt(F, f8, logn + 3, inverse=True)
    for i, c in enumerate(public_poly):
        f[i] = (f[i] + c) % p
    # divide by x^n - 1: q[i] = sum_{k >= 1} f[i + k n]; remainder must vanish
    q = [0] * (7 * n)
    for i in range(7 * n - 1, -1, -1):
        q[i] = (f[i + n] + (q[i + n] if i + n < 7 * n else 0)) % p
    for i in range(n):
        assert (f[i] + q[i]) % p == 0, "rest of division by vanishing polynomial"
    zm1 = list(z_poly); zm1[0] = (zm1[0] - 1) % p
    b1, r1 = _divide_by_linear(p, zm1, 1)
    b2, r2 = _divide_by_linear(p, zm1, cs["sid"][n - zk])
    assert r1 == 0 and r2 == 0, "Permutation: division rest"
    for i in range(len(b1)):
        q[i] = (q[i] + alphas[1] * b1[i] + alphas[2] * b2[i]) % p
    return _trim(q)


# ---------------------------------------------------------------------------------------------------- serde (rmp-serde layout)
def serialize_proof(curve: P.Curve, proof) -> bytes:
    """ProverProof as `rmp_serde::to_vec` writes it (proof.rs:51-195 with serde_as: structs are arrays in declaration order,
    points are ark-compressed 33-byte strings, field elements 32 bytes little-endian, Option::None is nil) -- the inverse of
    oracle/fixtures.py::load."""
    import msgpack
    fe = lambda x: int(x).to_bytes(32, "little")
    pt = lambda q: curve.compress(q)
    comm = lambda chunks: [[pt(c) for c in chunks]]
    pe = lambda e: None if e is None else [[fe(x) for x in e[0]], [fe(x) for x in e[1]]]
    ev = proof["evals"]
    lk = proof.get("lookup")
    srt = list(ev.get("lookup_sorted") or [])
    srt += [None] * (5 - len(srt))
    evals = [pe(ev["public"]), [pe(e) for e in ev["w"]], pe(ev["z"]), [pe(e) for e in ev["s"]], [pe(e) for e in ev["coefficients"]],
             pe(ev["generic_selector"]), pe(ev["poseidon_selector"]), pe(ev["complete_add_selector"]), pe(ev["mul_selector"]), pe(ev["emul_selector"]),
             pe(ev["endomul_scalar_selector"])] + [pe(e) for e in ev["optional_gate_selectors"]] + \
            [pe(ev.get("lookup_aggregation")), pe(ev.get("lookup_table")), [pe(e) for e in srt], pe(ev.get("runtime_lookup_table")),
             pe(ev.get("runtime_lookup_table_selector"))] + [pe((ev.get("lookup_selectors") or {}).get(q)) for q in K.LOOKUP_PATTERN_ORDER]
    op = proof["opening"]
    commitments = [[comm(c) for c in proof["w_comm"]], comm(proof["z_comm"]), comm(proof["t_comm"]),
                   None if lk is None else [[comm(c) for c in lk["sorted"]], comm(lk["aggreg"]), None if lk.get("runtime") is None else comm(lk["runtime"])]]
    opening = [[[pt(l), pt(r)] for l, r in op["lr"]], pt(op["delta"]), fe(op["z1"]), fe(op["z2"]), pt(op["sg"])]
    prev = [[[fe(x) for x in chals], comm(c)] for chals, c in proof.get("prev_challenges", [])]
    return msgpack.packb([commitments, opening, evals, fe(proof["ft_eval1"]), prev], use_bin_type=True)
