"""A complete Kimchi proof with every data-parallel step on the device -- the caller of the hot path, restated.

`ProverProof::create` (kimchi/src/prover.rs:187-1515) for circuits of generic gates (the benchmark circuit of
kimchi/src/bench.rs:59-122: 2^k - 10 `Const(1)` gates, 15 witness columns of ones), no lookups, no optional gates, no
recursion, one chunk (SRS size = domain size).  The columns never leave HBM between the witness upload and the opening
proof: witness -> 15 Lagrange-basis commitments -> iNTT -> permutation accumulator z (permutation.rs:447-577) -> commit
-> 8x extension of w, z (constraints.rs:487-507) -> generic-gate rows on d4 and permutation rows on d8 -> iNTT(4n),
iNTT(8n) -> division by Z_H with the remainder asserted ZERO -> boundary quotients -> 7-chunk commitment of t -> chunked
evaluations at zeta, zeta*omega -> ft -> combine_polys / b_init -> the 16 folding rounds of SRS::open.  The transcript runs
through the library's native sponges (kh_sponge_*), so every challenge is the real Fiat-Shamir value; the host keeps what
is scalar and sequential (challenge algebra, blinders, the Schnorr tail of the opening), as the reference does.

What is NOT evaluated: the constraints of Poseidon / CompleteAdd / VarBaseMul / EndoMul / EndoMulScalar.  The reference
evaluates them over d8 on every proof (prover.rs:824-868) although their selectors are the zero polynomial for such a
circuit; they contribute exactly zero to the quotient, so the proof is the same (their selector columns, commitments and
evaluations ARE part of the proof and of the transcript here).

The proof this produces is checked by the oracle's restatement of the reference VERIFIER (oracle/kimchi.py: Fiat-Shamir
replay with the oracle's own sponge, ft_eval0, ft_comm, SRS::verify) in tests/test_gpu_prover.py; bench.py times it.
This module is product code: it never imports the oracle."""
from __future__ import annotations

import hashlib
import time

import numpy as np

from . import khip
from . import polish as OP

COLUMNS, PERMUTS, ZK_ROWS = 15, 7, 3
ALPHA_PERM0 = 21                    # the gates register 21 powers of alpha first (linearization.rs:56-58), the permutation the next 3
MOD = {khip.FP: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
       khip.FQ: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}
R256 = 1 << 256


class Fld:
    """Host scalars: Python integers, converted to / from the wire form (4 Montgomery limbs) at the C ABI."""

    def __init__(self, fid: int):
        self.fid, self.p = fid, MOD[fid]
        self.rinv = pow(R256, -1, self.p)

    def limbs(self, v: int):
        return np.frombuffer(((v % self.p) * R256 % self.p).to_bytes(32, "little"), dtype=np.uint64).copy()

    def limbs_many(self, vs):
        if not len(vs):
            return np.zeros((0, 4), np.uint64)
        p = self.p
        return np.frombuffer(b"".join(((v % p) * R256 % p).to_bytes(32, "little") for v in vs), dtype=np.uint64).reshape(-1, 4).copy()

    def value(self, l) -> int:
        return int.from_bytes(np.ascontiguousarray(l, dtype=np.uint64).tobytes(), "little") * self.rinv % self.p

    def values(self, arr):
        """every element of an (..., 4) limb array, flattened, as integers"""
        b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
        return [int.from_bytes(b[i:i + 32], "little") * self.rinv % self.p for i in range(0, len(b), 32)]

    def inv(self, v: int) -> int:
        return pow(v, -1, self.p)

    def rand(self, rng) -> int:
        return int.from_bytes(rng.bytes(40), "little") % self.p

    def rand_many(self, rng, k: int):
        """k uniform field elements from ONE draw of the generator (the per-call overhead of numpy's Generator.bytes dominates `rand`)"""
        b = rng.bytes(40 * k)
        return [int.from_bytes(b[40 * i:40 * i + 40], "little") % self.p for i in range(k)]


def sample_shifts(F: Fld, log2_n: int):
    """Shifts::new (permutation.rs:140-199): shift_0 = 1, then quadratic non-residues outside the domain from Blake2b512(counter)."""
    n = 1 << log2_n
    counter = [7]

    def sample():
        while True:
            counter[0] += 1
            d = hashlib.blake2b(counter[0].to_bytes(4, "big"), digest_size=64).digest()
            s = int.from_bytes(d[:31], "little")
            if pow(s, (F.p - 1) // 2, F.p) == F.p - 1 and pow(s, n, F.p) != 1:
                return s
    shifts = [1]
    for _ in range(1, PERMUTS):
        s = sample()
        while s in shifts:
            s = sample()
        shifts.append(s)
    return shifts


def scalar_challenge(curve: int, F: Fld, chal: int) -> int:
    return F.value(khip.scalar_challenge_to_field(curve, chal))


class ProverIndex:
    """Index of a generic-gate circuit on a domain of 2^log2_n rows, device-resident: coefficient and selector columns,
    sigma, their coefficient forms and 8x extensions, x and the permutation vanishing polynomial on d8, the SRS with its
    Lagrange basis, and the verifier-index commitments + digest (verifier_index.rs:175-300, 405-500)."""

    GATE_TYPES = ("Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar")     # the always-present selectors after Generic

    def __init__(self, curve: int, log2_n: int, gate_coeffs, srs=None, gate_types=None, public: int = 0):
        """gate_coeffs: (rows, 15, 4) uint64 Montgomery limbs -- coefficient rows of the gates (rows <= n - 3).  gate_types: one name
        per row ("Generic", one of GATE_TYPES, or anything else -- "Zero", "Lookup" -- for a row without gate constraints); default: all Generic."""
        self.curve = curve
        self.public = public                                 # number of public inputs: witness[0][0..public] (constraints.rs:870-890)
        self.fid = khip.FP if curve == khip.VESTA else khip.FQ
        F = self.F = Fld(self.fid)
        self.log2_n, self.n = log2_n, 1 << log2_n
        n, fid = self.n, self.fid
        gate_coeffs = np.ascontiguousarray(gate_coeffs, dtype=np.uint64).reshape(-1, COLUMNS, 4)
        self.gates = gate_coeffs.shape[0]
        assert self.gates + ZK_ROWS <= n
        self.srs = srs if srs is not None else khip.Srs.create(curve, n)
        if self.srs.lagrange_chunks(log2_n) == 0:
            self.srs.compute_lagrange(log2_n)                 # SRS::lagrange_basis on the device (index time)
        self.h = khip.srs_h(curve)
        self.omega = F.value(khip.domain_generator(fid, log2_n))
        self.shifts = sample_shifts(F, log2_n)
        one = F.limbs(1)
        # ---- d1 evaluation columns: coefficients (15), generic selector, sid, sigma (7)
        co = np.zeros((COLUMNS, n, 4), dtype=np.uint64)
        co[:, :self.gates, :] = np.transpose(gate_coeffs, (1, 0, 2))
        self.gate_types = list(gate_types) if gate_types is not None else ["Generic"] * self.gates
        assert len(self.gate_types) == self.gates
        self.live_gate_types = set(self.gate_types)
        sel = np.zeros((n, 4), dtype=np.uint64)
        sel[[r for r, g in enumerate(self.gate_types) if g == "Generic"]] = one
        self.SEL0 = COLUMNS + 2 + PERMUTS                               # first of the five further selector columns
        self.d1 = khip.DevBuf((COLUMNS + 1 + 1 + PERMUTS + 5) * n * 32)  # [coef 0..14 | generic sel | sid | sigma 0..6 | psm add mul emul emulscalar]
        self.d1.upload_at(0, co); self.d1.upload_at(COLUMNS * n * 32, sel)
        for k, name in enumerate(self.GATE_TYPES):
            sk = np.zeros((n, 4), dtype=np.uint64)
            sk[[r for r, g in enumerate(self.gate_types) if g == name]] = one
            self.d1.upload_at((self.SEL0 + k) * n * 32, sk)
        xpoly = np.zeros((n, 4), dtype=np.uint64); xpoly[1] = one
        sid = self.col1(COLUMNS + 1)
        self.d1.upload_at((COLUMNS + 1) * n * 32, xpoly)
        khip.ntt_dev(fid, sid, log2_n, False, 1)                                                 # sid[j] = omega^j
        for i in range(PERMUTS):                                                                 # identity wiring: sigma_i = shift_i * sid
            khip.expr_evaluations_dev(fid, [OP.cell(0), (OP.TOK_CONST, 0), (OP.TOK_MUL, 0)], [sid], [n], F.limbs_many([self.shifts[i]]), n, self.col1(COLUMNS + 2 + i))
        self._finish_columns()

    # d1 / coefficient / d8 column views --------------------------------------------------------
    def col1(self, k):
        return self.d1.view(k * self.n * 32)

    def colc(self, k):
        return self.dc.view(k * self.n * 32)

    def col8(self, k):
        return self.d8.view(k * 8 * self.n * 32)

    def set_sigma(self, sigma_limbs):
        """Replace the identity wiring by sigma columns (7, n, 4) computed by the caller (copy constraints)."""
        self.d1.upload_at((COLUMNS + 2) * self.n * 32, np.ascontiguousarray(sigma_limbs, dtype=np.uint64))
        self._finish_columns()

    def _finish_columns(self):
        n, fid, F, logn = self.n, self.fid, self.F, self.log2_n
        ncol = COLUMNS + 1 + 1 + PERMUTS + 5                            # same order as d1
        if not hasattr(self, "dc"):
            self.dc = khip.DevBuf((ncol + 2) * n * 32)                  # + [x | zkpm] in coefficient form
            self.d8 = khip.DevBuf((ncol + 2) * 8 * n * 32)
            self.zero_poly = khip.DevBuf(n * 32).zero()                 # the five absent selectors: the zero polynomial
        khip.dev_copy(self.dc.ptr, self.d1.ptr, ncol * n * 32)
        khip.ntt_dev(fid, self.dc, logn, True, ncol)                    # coefficient forms
        xpoly = np.zeros((n, 4), dtype=np.uint64); xpoly[1] = F.limbs(1)
        a = pow(self.omega, n - ZK_ROWS, F.p); b = a * self.omega % F.p; c = pow(self.omega, n - 1, F.p)
        zk = np.zeros((n, 4), dtype=np.uint64)                          # (x - a)(x - b)(x - c): permutation_vanishing_polynomial
        zk[:4] = F.limbs_many([(-a * b * c) % F.p, (a * b + a * c + b * c) % F.p, (-(a + b + c)) % F.p, 1])
        self.zkpm_coeffs = [(-a * b * c) % F.p, (a * b + a * c + b * c) % F.p, (-(a + b + c)) % F.p, 1]
        self.dc.upload_at(ncol * n * 32, xpoly); self.dc.upload_at((ncol + 1) * n * 32, zk)
        khip.lde_dev(fid, self.dc, logn, 3, self.d8, ncol + 2)          # everything on d8
        self.X8, self.ZKPM8 = ncol, ncol + 1
        # ---- verifier-index commitments (commit_evaluations_non_hiding over the Lagrange basis; selectors masked with 1)
        com, inf = self.srs.msm_batch_dev(self.d1.ptr, n, COLUMNS + 1, basis=logn)
        self.coefficients_comm = [(com[i], bool(inf[i])) for i in range(COLUMNS)]
        g, ginf = self.srs.mask_custom(com[COLUMNS:COLUMNS + 1], inf[COLUMNS:COLUMNS + 1], F.limbs_many([1]))
        self.generic_comm = (g[0], bool(ginf[0]))
        com, inf = self.srs.msm_batch_dev(self.col1(COLUMNS + 2).ptr, n, PERMUTS, basis=logn)
        self.sigma_comm = [(com[i], bool(inf[i])) for i in range(PERMUTS)]
        self.zero_selector_comm = (self.h.copy(), False)                # identity + 1 * h
        com, inf = self.srs.msm_batch_dev(self.col1(self.SEL0).ptr, n, 5, basis=logn)
        com, inf = self.srs.mask_custom(com, inf, F.limbs_many([1] * 5))
        self.selector_comms = [(com[i], bool(inf[i])) for i in range(5)]  # psm, complete_add, mul, emul, endomul_scalar (= h for an absent gate type)
        sp = khip.Sponge(khip.Sponge.FQ, self.curve)
        for c_, i_ in self.sigma_comm + self.coefficients_comm + [self.generic_comm] + self.selector_comms:
            sp.absorb_g(c_.reshape(1, 8), np.array([1 if i_ else 0], dtype=np.uint8))
        self.digest = sp.squeeze_field()                                # VerifierIndex::digest -> digest_fq
        sp.free()
        khip.sync()

    def attach_lookup(self, LI):
        """Adds a lookup constraint system (proof_systems_amd.lookup.LookupIndex over the same domain): coefficient forms and
        8x extensions of the pattern selectors, the row-set atoms on d8, and the verifier-index side -- table columns and the
        table-id column committed and masked with 1, selectors committed non-hiding (verifier_index.rs:189-216) -- folded into
        the digest (verifier_index.rs:482-530)."""
        from . import lookup as LK
        assert LI.n == self.n and LI.fid == self.fid
        n, fid, F, logn = self.n, self.fid, self.F, self.log2_n
        self.lookup = LI
        LI.sel_c = {}; LI.sel8 = {}
        for q in LI.patterns:
            c = khip.DevBuf(n * 32); khip.dev_copy(c.ptr, LI.d_selectors[q].ptr, n * 32); khip.ntt_dev(fid, c, logn, True, 1)
            e = khip.DevBuf(8 * n * 32); khip.lde_dev(fid, c, logn, 3, e, 1)
            LI.sel_c[q], LI.sel8[q] = c, e
        LI.atoms8 = LK.atom_columns(LI, 3)
        one = F.limbs_many([1])

        def commit(vals, masked):
            com, inf = self.srs.commit_evaluations_non_hiding(logn, F.limbs_many(vals))
            if masked:
                com, inf = self.srs.mask_custom(com, inf, one)
            return (com[0], bool(inf[0]))
        LI.table_comm = [commit(c, True) for c in LI.table_cols]
        LI.table_ids_comm = commit(LI.table_ids, True) if LI.table_ids is not None else None
        LI.selector_comm = {q: commit(LI.selectors[q], False) for q in LI.patterns}
        sp = khip.Sponge(khip.Sponge.FQ, self.curve)
        extra = LI.table_comm + ([LI.table_ids_comm] if LI.table_ids_comm else []) + [LI.selector_comm[q] for q in LI.patterns]
        for c_, i_ in self.sigma_comm + self.coefficients_comm + [self.generic_comm] + self.selector_comms + extra:
            sp.absorb_g(c_.reshape(1, 8), np.array([1 if i_ else 0], dtype=np.uint8))
        self.digest = sp.squeeze_field()
        sp.free()
        khip.sync()

    def free(self):
        for b in (self.d1, self.dc, self.d8, self.zero_poly):
            b.free()


def bench_circuit_index(curve: int, log2_n: int, srs=None) -> ProverIndex:
    """BenchmarkCtx::new(log2_n) (kimchi/src/bench.rs:59-96): 2^log2_n - 10 generic gates `Const(1)`, wired to themselves."""
    fid = khip.FP if curve == khip.VESTA else khip.FQ
    F = Fld(fid)
    rows = (1 << log2_n) - 10
    co = np.zeros((rows, COLUMNS, 4), dtype=np.uint64)
    co[:, 0, :] = F.limbs(1); co[:, 4, :] = F.limbs(F.p - 1)           # 1 * w0 - 1 = 0
    return ProverIndex(curve, log2_n, co, srs)


def create_proof(ix: ProverIndex, witness, rng, timings=None, check: bool = True, witness_on_device=None):
    """ProverProof::create.  witness: (15, rows, 4) Montgomery limbs, rows <= n - 3 (padded with zeros, the last 3 rows
    randomised, prover.rs:254-266) -- or witness_on_device: a DevBuf already holding the padded (15, n, 4) columns.
    rng: numpy Generator (blinders, zero-knowledge rows).  Returns the proof as a dict of limb arrays / Python ints."""
    F, fid, n, logn, curve, srs = ix.F, ix.fid, ix.n, ix.log2_n, ix.curve, ix.srs
    t_start = time.perf_counter()
    marks = []

    def mark(name):
        marks.append((name, time.perf_counter()))
    one = F.limbs(1)
    NB = n * 32
    # ---- witness on the device: [w 0..14 | z] in evaluation form, then in coefficient form, then on d8
    ev = khip.DevBuf(16 * NB)
    if witness_on_device is None:
        # straight from the caller's columns (no padded host copy: that staging was 1.7 of this phase's 2.4 ms): zero the device
        # buffer, upload each column's rows, then the three zero-knowledge rows per column (prover.rs:254-286)
        wit = np.asarray(witness, dtype=np.uint64).reshape(COLUMNS, -1, 4)
        assert wit.shape[1] + ZK_ROWS <= n, "NoRoomForZkInWitness"
        if wit.shape[1] + ZK_ROWS < n:
            ev.zero()
        zk = F.limbs_many(F.rand_many(rng, COLUMNS * ZK_ROWS)).reshape(COLUMNS, ZK_ROWS, 4)
        ev.upload_2d(0, NB, wit)                             # 15 columns, each into its padded device column, one transfer
        ev.upload_2d((n - ZK_ROWS) * 32, NB, zk)
    else:
        khip.dev_copy(ev.ptr, witness_on_device.ptr, COLUMNS * NB)
    mark("witness_upload")
    fq = khip.Sponge(khip.Sponge.FQ, curve)
    fq.absorb(ix.digest)
    pub_c = None
    if ix.public:                                           # the negated public-input polynomial (prover.rs:281-309): -p_i on the first rows
        pe_ = np.zeros((n, 4), dtype=np.uint64)
        pub_vals = F.values(ev.download_at(0, (ix.public, 4)))
        pe_[:ix.public] = F.limbs_many([(-x) % F.p for x in pub_vals])
        pub_c = khip.DevBuf(NB).upload(pe_)
        com, inf = srs.msm_batch_dev(pub_c.ptr, n, 1, basis=logn)
        com, inf = srs.mask_custom(com, inf, F.limbs_many([1]))
        fq.absorb_g(com, inf)
        khip.ntt_dev(fid, pub_c, logn, True, 1)
    else:
        fq.absorb_g(ix.h.copy().reshape(1, 8))              # zero public polynomial: commit_non_hiding -> [0], masked with 1 -> h
    # ---- witness commitments (commit_evaluations_non_hiding x 15 in one batched MSM over the Lagrange basis) + blinders
    tk = srs.msm_submit(ev.ptr, n, COLUMNS, basis=logn)       # ... while the columns are interpolated on the main stream (both only read `ev`)
    cf = khip.DevBuf(16 * NB)                               # coefficient forms [w | z]
    khip.dev_copy(cf.ptr, ev.ptr, COLUMNS * NB)
    khip.ntt_dev(fid, cf, logn, True, COLUMNS)
    # The 8x extension of the witness needs no challenge either: queued here, it runs while the host blinds and absorbs the commitments
    # (~0.4 ms in which the main stream had nothing to do).  Generic gates + permutation read only w0..w6 (and z) on d8
    # (generic.rs:83-120, permutation.rs:216-331): half of the reference's 16 extensions.
    LI = getattr(ix, "lookup", None)
    e8 = khip.DevBuf(16 * 8 * NB)
    N8 = 8 * NB
    w8 = PERMUTS if LI is None and not (ix.live_gate_types & set(ix.GATE_TYPES)) else COLUMNS
    khip.lde_dev(fid, cf, logn, 3, e8, w8)
    com, inf = srs.msm_wait(tk)
    w_blind = F.rand_many(rng, COLUMNS)
    w_comm, w_inf = srs.mask_custom(com, inf, F.limbs_many(w_blind))
    fq.absorb_g(w_comm, w_inf)
    lkp = None
    if LI is not None:                                      # prover.rs:383-633: joint combiner, combined table, sorted columns
        from . import lookup as LK
        jc = scalar_challenge(curve, F, fq.challenge() if LI.joint_lookup_used else 0)
        d_table = LI.joint_table_dev(jc)
        table_ints = F.values(d_table.download((n, 4)))
        wcols = ev.download((COLUMNS, n, 4))
        used = sorted({c for q in LI.patterns for tid, entry in OP.LOOKUP_PATTERNS[q] for c in (list(entry) + ([tid[1]] if isinstance(tid, tuple) else []))})
        wit_ints = [F.values(wcols[c]) if c in used else None for c in range(COLUMNS)]
        srt = [LK.zk_patch(F, c, n, ZK_ROWS, rng) for c in LK.sorted_columns(LI, wit_ints, table_ints, jc)]    # ValueError(row): value not in the table
        d_sorted = [khip.DevBuf(NB).upload(F.limbs_many(c)) for c in srt]
        s_blind, s_comm = [], []
        for b in d_sorted:
            com, inf = srs.msm_batch_dev(b.ptr, n, 1, basis=logn)
            bl_ = F.rand(rng)
            com, inf = srs.mask_custom(com, inf, F.limbs_many([bl_]))
            fq.absorb_g(com, inf)
            s_blind.append(bl_); s_comm.append((com[0], bool(inf[0])))
        lkp = {"jc": jc, "d_table": d_table, "d_sorted": d_sorted, "s_blind": s_blind, "s_comm": s_comm}
    mark("witness_commit")
    beta = F.value(fq.challenge_field()); gamma = F.value(fq.challenge_field())
    if lkp is not None:                                     # prover.rs:635-673: the lookup aggregation, committed before z
        d_agg = LK.aggregation_dev(LI, [ev.view(i * NB) for i in range(COLUMNS)], lkp["d_sorted"], lkp["d_table"], lkp["jc"], beta, gamma, rng)
        if check and F.value(d_agg.download_at((n - ZK_ROWS - 1) * 32, (4,))) != 1:
            raise RuntimeError("final value of the lookup aggregation is not 1 (lookup/constraints.rs:325-331)")
        com, inf = srs.msm_batch_dev(d_agg.ptr, n, 1, basis=logn)
        a_blind = F.rand(rng)
        com, inf = srs.mask_custom(com, inf, F.limbs_many([a_blind]))
        fq.absorb_g(com, inf)
        lkp.update({"d_agg": d_agg, "a_blind": a_blind, "a_comm": (com[0], bool(inf[0]))})
    # ---- permutation accumulator z (perm_aggreg): numerators / denominators, batch inversion, running product
    d1cols = [ev.view(i * NB) for i in range(PERMUTS)] + [ix.col1(COLUMNS + 2 + i) for i in range(PERMUTS)] + [ix.col1(COLUMNS + 1)]
    consts = F.limbs_many([gamma, beta] + [beta * s % F.p for s in ix.shifts])
    num_t, den_t = OP.perm_aggreg_tokens()
    num = khip.DevBuf(NB); den = khip.DevBuf(NB)
    num.upload_at(0, one); den.upload_at(0, one)
    khip.expr_evaluations_dev(fid, num_t, d1cols, [n] * 15, consts, n - 1, num, out_offset=1)
    khip.expr_evaluations_dev(fid, den_t, d1cols, [n] * 15, consts, n - 1, den, out_offset=1)
    khip.batch_inversion_dev(fid, den, n - 1, offset=1)
    zcol = ev.view(COLUMNS * NB)
    khip.expr_evaluations_dev(fid, [OP.cell(0), OP.cell(1), (OP.TOK_MUL, 0)], [num, den], [n, n], one.reshape(1, 4), n, zcol)
    khip.field_scan_dev(fid, khip.SCAN_MUL, zcol, n - ZK_ROWS + 1)
    if check:
        last = F.value(ev.download_at((COLUMNS * n + n - ZK_ROWS) * 32, (4,)))
        if last != 1:
            raise RuntimeError("final value of the permutation accumulator is not 1 (permutation.rs:566-568)")
    ev.upload_at((COLUMNS * n + n - ZK_ROWS + 1) * 32, F.limbs_many(F.rand_many(rng, 2)))     # the two random rows
    khip.dev_copy(cf.ptr + COLUMNS * NB, zcol.ptr, NB)
    khip.ntt_dev(fid, cf.view(COLUMNS * NB), logn, True, 1)
    zc = cf.view(COLUMNS * NB)
    tk = srs.msm_submit(zc.ptr, n, 1)                       # the commitment to z runs while z is extended to d8
    khip.lde_dev(fid, zc, logn, 3, e8.view(COLUMNS * N8), 1)
    com, inf = srs.msm_wait(tk)
    z_blind = F.rand(rng)
    z_comm, z_inf = srs.mask_custom(com, inf, F.limbs_many([z_blind]))
    fq.absorb_g(z_comm, z_inf)
    mark("z")
    alpha = scalar_challenge(curve, F, fq.challenge())
    alphas = [pow(alpha, ALPHA_PERM0 + i, F.p) for i in range(3)]
    # ---- constraint rows on d8, quotient
    gen_cols = [e8.view(i * N8) for i in range(6)] + [ix.col8(i) for i in range(10)] + [ix.col8(COLUMNS)]
    t4 = khip.DevBuf(4 * NB); t8 = khip.DevBuf(N8)
    khip.expr_evaluations_dev(fid, OP.generic_gate_tokens(0, 6, 16, 0, 1), gen_cols, [8 * n] * 17, F.limbs_many([1, alpha]), 4 * n, t4, stride=2, next_shift=8)
    perm_cols = [e8.view(i * N8) for i in range(PERMUTS)] + [ix.col8(COLUMNS + 2 + i) for i in range(PERMUTS)] + [e8.view(COLUMNS * N8), ix.col8(ix.X8), ix.col8(ix.ZKPM8)]
    pconsts = F.limbs_many([gamma, beta, alphas[0]] + [beta * s % F.p for s in ix.shifts])
    khip.expr_evaluations_dev(fid, OP.perm_quot_tokens(w0=0, s0=7, z=14, x=15, zkpm=16, gamma=0, beta=1, bshift0=3, alpha0=2), perm_cols, [8 * n] * 17, pconsts,
                              8 * n, t8, stride=1, next_shift=8)
    live_gates = [(k_, name) for k_, name in enumerate(ix.GATE_TYPES) if name in ix.live_gate_types]
    if live_gates:                                          # the gate library on d8 (prover.rs:824-868): index(gate) * sum_i alpha^i constraint_i
        endo_q = F.value(khip.endos(1 - curve)[0])          # VerifierIndex::endo = endos::<OtherCurve>().0, an element of this scalar field
        gcols = [e8.view(i * N8) for i in range(COLUMNS)] + [ix.col8(i) for i in range(COLUMNS)]
        for k_, name in live_gates:
            gtoks, gconsts = OP.gate_program(name, F.p, alpha, selector_col=30, mds=OP.POSEIDON_MDS[fid], endo=endo_q)
            khip.expr_evaluations_dev(fid, gtoks, gcols + [ix.col8(ix.SEL0 + k_)], [8 * n] * 31, F.limbs_many(gconsts), 8 * n, t8, stride=1, next_shift=8, accumulate=True)
    if lkp is not None:                                     # the lookup constraints on d8 (prover.rs:874-903), powers alpha^24 ...
        nl = len(lkp["d_sorted"]) + 2
        lkc = khip.DevBuf(nl * NB); lk8 = khip.DevBuf(nl * N8)   # coefficient forms / d8: [sorted ... | aggregation | combined table]
        for k_, b in enumerate(lkp["d_sorted"] + [lkp["d_agg"], lkp["d_table"]]):
            khip.dev_copy(lkc.ptr + k_ * NB, b.ptr, NB)
        khip.ntt_dev(fid, lkc, logn, True, nl)
        khip.lde_dev(fid, lkc, logn, 3, lk8, nl)
        cols = LK.column_layout(LI)
        _, tic_expr = LI.constraint_combiners(lkp["jc"])
        ltoks, lconsts = OP.lookup_program(F.p, LI.patterns, cols, lkp["jc"], tic_expr, beta, gamma, alpha, alpha0=ALPHA_PERM0 + 3)
        lbufs = [e8.view(i * N8) for i in range(COLUMNS)] + [lk8.view(k_ * N8) for k_ in range(nl)] + [LI.sel8[q] for q in LI.patterns] + list(LI.atoms8)
        assert len(lbufs) == cols["count"]
        khip.expr_evaluations_dev(fid, ltoks, lbufs, [8 * n] * len(lbufs), F.limbs_many(lconsts), 8 * n, t8, stride=1, next_shift=8, accumulate=True)
        lkp.update({"lkc": lkc, "lk8": lk8, "nl": nl})
    khip.ntt_dev(fid, t4, logn + 2, True, 1)
    khip.ntt_dev(fid, t8, logn + 3, True, 1)
    if pub_c is not None:
        khip.poly_lincomb_dev(fid, [t8, t4, pub_c], [8 * n, 4 * n, n], F.limbs_many([1, 1, 1]), t8, 8 * n)    # f = t4 + t8 + public (prover.rs:906-908)
    else:
        khip.poly_lincomb_dev(fid, [t8, t4], [8 * n, 4 * n], F.limbs_many([1, 1]), t8, 8 * n)
    quot = khip.DevBuf(7 * NB); rem = khip.DevBuf(NB)
    khip.divide_by_vanishing_poly_dev(fid, t8, 8 * n, logn, quot, rem)
    if check and rem.download((n, 4)).any():
        raise RuntimeError("rest of division by vanishing polynomial (prover.rs:913-917)")
    zm1 = khip.DevBuf(NB); b1 = khip.DevBuf(NB); b2 = khip.DevBuf(NB)
    khip.poly_lincomb_dev(fid, [zc], [n], F.limbs_many([1]), zm1, n)
    z0 = F.value(zm1.download_at(0, (4,)))
    zm1.upload_at(0, F.limbs((z0 - 1) % F.p))
    b1.zero(); b2.zero()
    for a_, dst in ((1, b1), (pow(ix.omega, n - ZK_ROWS, F.p), b2)):
        r_ = khip.divide_by_linear_dev(fid, zm1, n, F.limbs(a_), dst)
        if check and r_.any():
            raise RuntimeError("permutation boundary division rest (permutation.rs:301-321)")
    khip.poly_lincomb_dev(fid, [quot, b1, b2], [7 * n, n - 1, n - 1], F.limbs_many([1, alphas[1], alphas[2]]), quot, 7 * n)
    com, inf = srs.msm_batch_dev(quot.ptr, n, 7)
    t_blind = F.rand_many(rng, 7)
    t_comm, t_inf = srs.mask_custom(com, inf, F.limbs_many(t_blind))
    fq.absorb_g(t_comm, t_inf)
    mark("quotient")
    zeta = scalar_challenge(curve, F, fq.challenge())
    zetaw = zeta * ix.omega % F.p
    fq_before = fq.clone()
    # ---- evaluations at zeta, zeta * omega (coefficient forms; one chunk each)
    polys = [zc, ix.colc(COLUMNS)] + [ix.colc(ix.SEL0 + k_) for k_ in range(5)] + [cf.view(i * NB) for i in range(COLUMNS)] + [ix.colc(i) for i in range(COLUMNS)] + \
            [ix.colc(COLUMNS + 2 + i) for i in range(PERMUTS - 1)]
    lk_polys = []
    if lkp is not None:                                     # opening order (prover.rs:1368-1420): sorted ..., aggregation, combined table, pattern selectors
        lk_polys = [lkp["lkc"].view(k_ * NB) for k_ in range(lkp["nl"])] + [LI.sel_c[q] for q in LI.patterns]
    pts = F.limbs_many([zeta, zetaw])
    evl = khip.evaluate_chunks_batch_dev(fid, polys + lk_polys, [n] * (len(polys) + len(lk_polys)), [1] * (len(polys) + len(lk_polys)), n, pts)
    E = [tuple(F.values(e)) for e in evl]                 # one chunk per polynomial: (value at zeta, value at zeta * omega)
    pub_eval = (0, 0)
    if pub_c is not None:
        pe2 = khip.evaluate_chunks_dev(fid, pub_c, n, n, 1, pts)
        pub_eval = (F.value(pe2[0, 0]), F.value(pe2[1, 0]))
    evals = {"public": pub_eval, "z": E[0], "generic_selector": E[1], "poseidon_selector": E[2], "complete_add_selector": E[3], "mul_selector": E[4],
             "emul_selector": E[5], "endomul_scalar_selector": E[6], "w": E[7:22], "coefficients": E[22:37], "s": E[37:43]}
    lk_evals_open, lk_evals_sponge = [], []
    if lkp is not None:
        ns = len(lkp["d_sorted"])
        evals["lookup_sorted"] = E[43:43 + ns]; evals["lookup_aggregation"] = E[43 + ns]; evals["lookup_table"] = E[44 + ns]
        evals["lookup_selectors"] = {q: E[45 + ns + k_] for k_, q in enumerate(LI.patterns)}
        lk_evals_open = E[43:]
        lk_evals_sponge = [evals["lookup_aggregation"], evals["lookup_table"]] + list(evals["lookup_sorted"]) + [evals["lookup_selectors"][q] for q in LI.patterns]
    # ---- ft = perm_scalar * sigma_6 - (zeta^n - 1) * sum_i zeta^(n i) t_i   (Maller; prover.rs:1147-1188)
    zeta1 = pow(zeta, n, F.p)
    zkp = (zeta - pow(ix.omega, n - 3, F.p)) * (zeta - pow(ix.omega, n - 2, F.p)) % F.p * (zeta - pow(ix.omega, n - 1, F.p)) % F.p
    scal = evals["z"][1] * beta % F.p * alphas[0] % F.p * zkp % F.p
    for w_, s_ in zip(evals["w"], evals["s"]):
        scal = scal * ((gamma + beta * s_[0] + w_[0]) % F.p) % F.p
    scal = (-scal) % F.p
    ft = khip.DevBuf(NB)
    m1 = (-(zeta1 - 1)) % F.p
    khip.poly_lincomb_dev(fid, [ix.colc(COLUMNS + 2 + PERMUTS - 1)] + [quot.view(i * NB) for i in range(7)], [n] * 8,
                          F.limbs_many([scal] + [m1 * pow(zeta1, i, F.p) % F.p for i in range(7)]), ft, n)
    fte = khip.evaluate_chunks_dev(fid, ft, n, n, 1, pts)
    ft_eval0, ft_eval1 = F.value(fte[0, 0]), F.value(fte[1, 0])
    blinding_ft = m1 * sum(b * pow(zeta1, i, F.p) for i, b in enumerate(t_blind)) % F.p
    # ---- Fr-sponge: v, u
    fr = khip.Sponge(khip.Sponge.FR, curve)
    fr.absorb(fq.digest())
    empty = khip.Sponge(khip.Sponge.FR, curve); fr.absorb(empty.digest()); empty.free()
    order = [evals["z"], evals["generic_selector"], evals["poseidon_selector"], evals["complete_add_selector"], evals["mul_selector"], evals["emul_selector"],
             evals["endomul_scalar_selector"]] + list(evals["w"]) + list(evals["coefficients"]) + list(evals["s"])
    flat = [ft_eval1, pub_eval[0], pub_eval[1]] + [x for e in order + lk_evals_sponge for x in e]       # plonk_sponge.rs:92-155
    fr.absorb(F.limbs_many(flat))
    v = scalar_challenge(curve, F, fr.challenge())
    u = scalar_challenge(curve, F, fr.challenge())
    fr.free()
    mark("evaluations")
    # ---- SRS::open on (public, ft, z, 6 selectors, w x 15, coefficients x 15, sigma x 6)
    open_polys = [pub_c if pub_c is not None else ix.zero_poly, ft] + polys + lk_polys
    open_lens = [n if pub_c is not None else 0, n] + [n] * (len(polys) + len(lk_polys))
    blinders = [1, blinding_ft, z_blind, 1, 1, 1, 1, 1, 1] + w_blind + [0] * COLUMNS + [0] * (PERMUTS - 1)
    if lkp is not None:                                     # the combined table's blinder: sum_i jc^i over its masked columns + the table-id combiner (prover.rs:1384-1400)
        jc_, tic_ = LI.combiners(lkp["jc"])
        tb = sum(pow(jc_, i, F.p) for i in range(len(LI.table_cols))) + tic_
        blinders += lkp["s_blind"] + [lkp["a_blind"], tb % F.p] + [0] * len(LI.patterns)
    all_evals = [pub_eval, (ft_eval0, ft_eval1)] + order + lk_evals_open
    a_dev = khip.DevBuf(NB); b_dev = khip.DevBuf(NB)
    khip.combine_polys_dev(fid, open_polys, open_lens, [1] * len(open_polys), F.limbs(v), n, a_dev)
    khip.b_init_dev(fid, pts, F.limbs(u), n, b_dev)
    blinding_factor, cip, ps = 0, 0, 1
    for bl, (e0, e1) in zip(blinders, all_evals):
        blinding_factor = (blinding_factor + bl * ps) % F.p
        cip = (cip + ps * ((e0 + u * e1) % F.p)) % F.p                  # combined_inner_product (commitment.rs:622-657) = <p, b_init>
        ps = ps * v % F.p
    sp = fq_before
    bl = F.rand_many(rng, 2 * logn + 2)         # the reference's draw order: (rand_l, rand_r) per round, then d, r_delta
    lr_xy, lr_inf, delta, dinf, z1_l, z2_l, sg, sg_inf = khip.ipa_open(srs, a_dev, b_dev, n, F.limbs(cip), F.limbs(blinding_factor), sp, F.limbs_many(bl))
    opening = {"lr": [(lr_xy[r], lr_inf[r]) for r in range(logn)], "delta": (delta, dinf), "z1": F.value(z1_l), "z2": F.value(z2_l), "sg": (sg, sg_inf)}
    sp.free(); fq.free()
    mark("opening")
    for b in (ev, cf, e8, t4, t8, quot, rem, zm1, b1, b2, ft, a_dev, b_dev, num, den):
        b.free()
    if pub_c is not None:
        pub_c.free()
    lk_out = {}
    if lkp is not None:
        lk_out = {"lookup": {"sorted": lkp["s_comm"], "aggreg": lkp["a_comm"]}}
        for b in lkp["d_sorted"] + [lkp["d_agg"], lkp["d_table"], lkp["lkc"], lkp["lk8"]]:
            b.free()
    if timings is not None:
        prev = t_start
        for name, t in marks:
            timings[name] = timings.get(name, 0.0) + (t - prev); prev = t
        timings["total"] = timings.get("total", 0.0) + (marks[-1][1] - t_start)
    return {"w_comm": (w_comm, w_inf), "z_comm": (z_comm, z_inf), "t_comm": (t_comm, t_inf), "evals": evals, "ft_eval1": ft_eval1, "opening": opening,
            "challenges": {"beta": beta, "gamma": gamma, "alpha": alpha, "zeta": zeta, "v": v, "u": u}, **lk_out}
