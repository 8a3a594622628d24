"""TEST INFRASTRUCTURE ONLY (see oracle/pasta.py): kimchi's circuit description -- gates with wires and coefficients -- and
`ConstraintSystem::create(gates).build()` restated, so that the circuits of the reference's OWN tests (whose proofs /
verifier indexes are stored as golden vectors) can be rebuilt from their definition: gadgets, wiring, domain size,
zero-knowledge rows, sigma, coefficient and selector columns, the lookup constraint system.

  CircuitGate / Wire / connect_cell_pair   kimchi/src/circuits/gate.rs:150-200, 371-376; wires.rs:18-60
  create_generic_gadget                    circuits/polynomials/generic.rs:158-231
  extend_xor_gadget / create_xor_gadget    circuits/polynomials/xor.rs:41-89
  extend_and / create_and                  circuits/polynomials/and.rs:80-138
  ConstraintSystem::build                  circuits/constraints.rs:883-1100 (lookup domain size, zk_rows, padding, sigma)
  FeatureFlags::from_gates                 circuits/constraints.rs:770-830
  RandomField::gen                         utils/src/field_helpers.rs:42-54 (num-bigint 0.4 RandBigInt::gen_biguint_below on StdRng)

Pinned: the verifier index rebuilt from `extend_and(8)` equals, commitment for commitment, the index the reference serialised
in `and_prove_and_verify_vesta.bin` (tests/test_reference_kat.py)."""
from typing import Dict, List, Optional, Sequence, Tuple

from . import kimchi as K
from . import lookup as L
from . import pasta as P

COLUMNS, PERMUTS = 15, 7
OPTIONAL_GATES = K.OPTIONAL_GATES


# ------------------------------------------------------------------------------------------------------ gates
def gate(typ: str, row: int, coeffs: Sequence[int] = ()):
    """CircuitGate::new(typ, Wire::for_row(row), coeffs): wires[col] = (row, col)."""
    return {"typ": typ, "wires": [(row, c) for c in range(PERMUTS)], "coeffs": list(coeffs)}


def connect_cell_pair(gates, cell_pre: Tuple[int, int], cell_new: Tuple[int, int]):
    """gate.rs:372-376 -- cells are (row, column)."""
    tmp = gates[cell_pre[0]]["wires"][cell_pre[1]]
    gates[cell_pre[0]]["wires"][cell_pre[1]] = gates[cell_new[0]]["wires"][cell_new[1]]
    gates[cell_new[0]]["wires"][cell_new[1]] = tmp


def generic_spec(p: int, kind: str, **kw) -> List[int]:
    """The five coefficients of one half of a double generic gate (generic.rs:165-193); None = the reference's default."""
    g = lambda k, d: kw[k] % p if kw.get(k) is not None else d % p
    if kind == "Add":
        return [g("left", 1), g("right", 1), g("output", -1), 0, 0]
    if kind == "Mul":
        return [0, 0, g("output", -1), g("mul", 1), 0]
    if kind == "Const":
        return [1, 0, 0, 0, (-kw["cst"]) % p]
    if kind == "Pub":
        return [1, 0, 0, 0, 0]
    if kind == "Plus":
        return [1, 0, p - 1, 0, kw["cst"] % p]
    raise ValueError(kind)


def generic_gadget(p: int, row: int, spec1: List[int], spec2: Optional[List[int]] = None):
    return gate("Generic", row, list(spec1) + list(spec2 if spec2 is not None else [0] * 5))


def extend_xor_gadget(p: int, gates, bits: int) -> int:
    new_row = len(gates)
    nx = -(-bits // 16)
    for i in range(nx):
        gates.append(gate("Xor16", new_row + i))
    gates.append(generic_gadget(p, new_row + nx, generic_spec(p, "Const", cst=0)))
    zero_row = len(gates) - 1
    connect_cell_pair(gates, (zero_row, 0), (zero_row, 1))
    connect_cell_pair(gates, (zero_row, 0), (zero_row, 2))
    return len(gates)


def extend_and(p: int, gates, nbytes: int) -> int:
    xor_row = len(gates)
    and_row = extend_xor_gadget(p, gates, nbytes * 8)
    gates.append(generic_gadget(p, and_row, generic_spec(p, "Add"), generic_spec(p, "Add", right=-1, output=-2)))
    connect_cell_pair(gates, (xor_row, 0), (and_row, 0))
    connect_cell_pair(gates, (xor_row, 1), (and_row, 1))
    connect_cell_pair(gates, (and_row, 2), (and_row, 3))
    connect_cell_pair(gates, (xor_row, 2), (and_row, 4))
    return len(gates)


def gen_field_with_bits(rng: P.StdRng, bits: int) -> int:
    """StdRng::gen_biguint_below(2^bits) (num-bigint 0.4.x bigrand.rs): draw bound.bits() = bits + 1 bits as ceil((bits+1)/32)
    u32 words filled by `Rng::fill` (little-endian words off the block buffer), the last word shifted down to the remaining
    bit count; retry until the value is below the bound."""
    bit_size = bits + 1
    digits, rem = divmod(bit_size, 32)
    ln = digits + (1 if rem else 0)
    while True:
        words = [rng.next_u32() for _ in range(ln)]
        if rem:
            words[-1] >>= 32 - rem
        v = sum(w << (32 * i) for i, w in enumerate(words))
        if v < (1 << bits):
            return v


# ------------------------------------------------------------------------------------------------------ constraint system
def zk_rows_strict_lower_bound(num_chunks: int) -> int:
    return (2 * (PERMUTS + 1) * num_chunks - 2) // PERMUTS           # constraints.rs:769-771


GATE_TABLE = {"Xor": ("Xor", 256), "RangeCheck": ("RangeCheck", 4096), "ForeignFieldMul": ("RangeCheck", 4096), "Lookup": (None, 0)}


def build(F: P.Field, gates, public: int = 0, lookup_tables=(), max_poly_size: Optional[int] = None, prev_challenges: int = 0, runtime_tables=None):
    """ConstraintSystem::build + the column evaluations of the index on d1 (prover_index.rs / constraints.rs:596-731):
    returns a dict with n, zk_rows, omega, shifts, sid, coefficients[15], sigma[7], selectors {gate type -> column},
    optional (the enabled optional gate types), lookup (oracle.lookup.LookupCS or None), gate_types (per row)."""
    p = F.p
    gates = [dict(g, wires=list(g["wires"]), coeffs=list(g["coeffs"])) for g in gates]
    assert len(gates) > 1
    types = [g["typ"] for g in gates]
    info = L.LookupInfo(types, uses_runtime_tables=runtime_tables is not None)
    lookup_domain_size = sum(len(t["data"][0]) if t["data"] else 0 for t in lookup_tables)
    if runtime_tables is not None:
        lookup_domain_size += sum(len(rt["first_column"]) for rt in runtime_tables)
    tabs = {GATE_TABLE[q][0]: GATE_TABLE[q][1] for q in info.patterns if GATE_TABLE[q][0]}
    lookup_domain_size += sum(tabs.values())
    if not any(t["id"] == 0 for t in lookup_tables):
        lookup_domain_size += 1
    lower = max(len(gates), lookup_domain_size + 1)
    zk_rows = 3
    bound = lower + zk_rows
    if max_poly_size is not None:
        while True:
            size = 1 << (bound - 1).bit_length()
            num_chunks = 1 if size < max_poly_size else size // max_poly_size
            zk_rows = zk_rows_strict_lower_bound(num_chunks) + 1
            bound = lower + zk_rows
            if not size < bound:
                break
    log2_n = (bound - 1).bit_length()
    n = 1 << log2_n
    assert n > zk_rows
    for i in range(len(gates), n):
        gates.append(gate("Zero", i))
    types = [g["typ"] for g in gates]
    omega = F.root_of_unity(log2_n)
    sid = [1] * n
    for j in range(1, n):
        sid[j] = sid[j - 1] * omega % p
    shifts = K.sample_shifts(F, log2_n)
    coeffs = [[0] * n for _ in range(COLUMNS)]
    for r, g in enumerate(gates):
        for c, v in enumerate(g["coeffs"][:COLUMNS]):
            coeffs[c][r] = v % p
    sigma = [[shifts[g["wires"][c][1]] * sid[g["wires"][c][0]] % p for g in gates] for c in range(PERMUTS)]
    for row in range(n + 2 - zk_rows, n - 1):                          # constraints.rs:523-530: the sigmas of the zero-knowledge rows the permutation
        for c in range(PERMUTS):                                       # argument still checks (none for zk_rows = 3) are zero
            sigma[c][row] = 0
    sel = lambda names: [1 if t in names else 0 for t in types]
    selectors = {"Generic": sel(("Generic",)), "Poseidon": sel(("Poseidon",)), "CompleteAdd": sel(("CompleteAdd",)), "VarBaseMul": sel(("VarBaseMul",)),
                 "EndoMul": sel(("EndoMul",)), "EndoMulScalar": sel(("EndoMulScalar",))}
    optional = [t for t in OPTIONAL_GATES if t in types]
    for t in optional:
        selectors[t] = sel((t,))
    lcs = None
    if info.patterns:
        lcs = L.LookupCS(p, types, list(lookup_tables), n, zk_rows, runtime_tables=runtime_tables)
    return {"F": F, "log2_n": log2_n, "n": n, "zk_rows": zk_rows, "omega": omega, "sid": sid, "shifts": shifts, "coefficients": coeffs, "sigma": sigma,
            "selectors": selectors, "optional": optional, "lookup": lcs, "gate_types": types, "gates": gates, "public": public,
            "prev_challenges": prev_challenges, "generic_selector": selectors["Generic"]}


def verify_witness(cs, witness) -> None:
    """ProverIndex::verify for the gate types restated here (generic, Xor16 + the five library gates through oracle/gates.py)
    and the copy constraints: raises AssertionError on the first violated row."""
    from . import gates as G
    F = cs["F"]; p = F.p; n = cs["n"]
    rows = len(witness[0])
    w = [list(c) + [0] * (n - rows) for c in witness]
    for r, g in enumerate(cs["gates"]):
        for c in range(PERMUTS):
            r2, c2 = g["wires"][c]
            assert w[c][r] == w[c2][r2], ("copy constraint", r, c)
        curr = [w[c][r] for c in range(COLUMNS)]
        nxt = [w[c][(r + 1) % n] for c in range(COLUMNS)]
        if g["typ"] == "Generic":
            co = [cs["coefficients"][c][r] for c in range(COLUMNS)]
            pub = w[0][r] if r < cs["public"] else 0
            assert (co[0] * curr[0] + co[1] * curr[1] + co[2] * curr[2] + co[3] * curr[0] * curr[1] + co[4] - pub) % p == 0, ("generic", r)
            assert (co[5] * curr[3] + co[6] * curr[4] + co[7] * curr[5] + co[8] * curr[3] * curr[4] + co[9]) % p == 0, ("generic", r)
        elif g["typ"] in G.ROW_MACHINES:
            from . import poseidon as S
            co = [cs["coefficients"][c][r] for c in range(COLUMNS)]
            mds = S.params("fp" if F is P.Fp else "fq")["mds"]
            assert G.combined_row(F, g["typ"], curr, nxt, co, 7, mds=mds, endo=P.endos(P.PALLAS if F is P.Fp else P.VESTA)[0]) == 0, (g["typ"], r)
