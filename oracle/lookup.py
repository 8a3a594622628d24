"""TEST INFRASTRUCTURE ONLY -- CPU restatement of Kimchi's lookup argument (plookup with the snake-shaped sorted
columns), the part of the quotient the gate library does not cover.  Python integers mod p; nothing here is imported
by the product.  Parity is pinned BY DEFINITION only (the reference holds no golden vectors for these functions):
the module is checked against the reference's own checker `lookup::constraints::verify` (restated as `verify`), the
"final value is 1" assertion of `aggregation`, and the vanishing of every constraint row on a satisfied witness.

Follows (file:line in /root/reference/kimchi/src/circuits/lookup/):
  tables/xor.rs:9-30, tables/range_check.rs:10-22, tables/mod.rs:13,16 (ids), tables/mod.rs:147-162 (combine_table_entry)
  lookups.rs:417-520 (LookupPattern: lookups per row, joint sizes, tables, from_gate)
  lookups.rs:176-210 (LookupInfo::create), lookups.rs:222-264 (selectors), lookups.rs:266-280 (by_row)
  index.rs:188-430 (LookupConstraintSystem::create: concatenated table columns, table ids, padding)
  constraints.rs:35-48 (zk_patch), :90-194 (sorted), :233-338 (aggregation), :378-673 (constraints), :692-796 (verify)
  ../expr.rs:140-147 (unnormalized_lagrange_basis), ../polynomials/permutation.rs:78-89 (eval_vanishes_on_last_n_rows)
"""
from typing import Dict, List, Optional, Sequence, Tuple

XOR_TABLE_ID = 0
RANGE_CHECK_TABLE_ID = 1
CURR, NEXT = 0, 1


# ---------------------------------------------------------------- tables
def xor_table(p: int):
    """4-bit XOR table, reversed so that the LAST row is (0, 0, 0) (xor.rs:9-30)."""
    data = [[], [], []]
    for i in range(16):
        for j in range(16):
            data[0].append(i); data[1].append(j); data[2].append(i ^ j)
    for r in data:
        r.reverse()
        assert r[-1] == 0
    return {"id": XOR_TABLE_ID, "data": data}


def range_check_table(p: int):
    return {"id": RANGE_CHECK_TABLE_ID, "data": [list(range(1 << 12))]}


def combine_table_entry(p: int, joint_combiner: int, table_id_combiner: int, row: Sequence[int], table_id: int) -> int:
    acc = 0
    for x in reversed(list(row)):
        acc = (joint_combiner * acc + x) % p
    return (acc + table_id_combiner * table_id) % p


# ---------------------------------------------------------------- patterns
# a joint lookup spec: (table_id, entries); table_id = ("const", id) | ("wit", column);
# an entry (SingleLookup) = [(coefficient, (row, column)), ...]
def _l(col):
    return [(1, (CURR, col))]


PATTERNS = {
    "Xor": {"max_per_row": 4, "max_joint_size": 3, "table": "Xor",
            "lookups": [(("const", XOR_TABLE_ID), [_l(3 + i), _l(7 + i), _l(11 + i)]) for i in range(4)]},
    "Lookup": {"max_per_row": 3, "max_joint_size": 2, "table": None,
               "lookups": [(("wit", 0), [_l(2 * i + 1), _l(2 * i + 2)]) for i in range(3)]},
    "RangeCheck": {"max_per_row": 4, "max_joint_size": 1, "table": "RangeCheck",
                   "lookups": [(("const", RANGE_CHECK_TABLE_ID), [_l(c)]) for c in range(3, 7)]},
    "ForeignFieldMul": {"max_per_row": 4, "max_joint_size": 1, "table": "RangeCheck",
                        "lookups": [(("const", RANGE_CHECK_TABLE_ID), [_l(c)]) for c in range(7, 11)]},
}
PATTERN_ORDER = ["Xor", "Lookup", "RangeCheck", "ForeignFieldMul"]          # LookupPatterns::into_iter


def pattern_from_gate(typ: str, row: int) -> Optional[str]:
    """LookupPattern::from_gate (lookups.rs:500-513)."""
    if typ == "Lookup" and row == CURR:
        return "Lookup"
    if (typ == "RangeCheck0" and row == CURR) or typ == "RangeCheck1" or (typ == "Rot64" and row == CURR):
        return "RangeCheck"
    if typ == "ForeignFieldMul":
        return "ForeignFieldMul"
    if typ == "Xor16" and row == CURR:
        return "Xor"
    return None


class LookupInfo:
    """LookupInfo::create_from_gates (lookups.rs:176-206); `gates` = list of gate type names."""

    def __init__(self, gates: Sequence[str], uses_runtime_tables: bool = False):
        used = set()
        for g in gates:
            for r in (CURR, NEXT):
                pat = pattern_from_gate(g, r)
                if pat:
                    used.add(pat)
        self.patterns = [q for q in PATTERN_ORDER if q in used]
        self.max_per_row = max((PATTERNS[q]["max_per_row"] for q in self.patterns), default=0)
        self.max_joint_size = max((PATTERNS[q]["max_joint_size"] for q in self.patterns), default=0)
        self.joint_lookup_used = any(PATTERNS[q]["max_joint_size"] > 1 for q in self.patterns)
        self.uses_runtime_tables = uses_runtime_tables

    def by_row(self, gates: Sequence[str]):
        kinds = [[] for _ in range(len(gates) + 1)]
        for i, g in enumerate(gates):
            pat = pattern_from_gate(g, CURR)
            if pat:
                kinds[i] = PATTERNS[pat]["lookups"]
            pat = pattern_from_gate(g, NEXT)
            if pat:
                kinds[i + 1] = PATTERNS[pat]["lookups"]
        return kinds

    def pattern_by_row(self, gates: Sequence[str]):
        kinds = [None] * (len(gates) + 1)
        for i, g in enumerate(gates):
            pat = pattern_from_gate(g, CURR)
            if pat:
                kinds[i] = pat
            pat = pattern_from_gate(g, NEXT)
            if pat:
                kinds[i + 1] = pat
        return kinds


def spec_value(p: int, spec, joint_combiner: int, table_id_combiner: int, witness, i: int) -> int:
    """JointLookup::evaluate (lookups.rs:376-392) on the witness rows i / i + 1; witness[col][row]."""
    tid, entries = spec
    ev = lambda pos: witness[pos[1]][i + (1 if pos[0] == NEXT else 0)]
    table_id = tid[1] % p if tid[0] == "const" else ev((CURR, tid[1]))
    vals = [sum(c * ev(pos) for c, pos in e) % p for e in entries]
    return combine_table_entry(p, joint_combiner, table_id_combiner, vals, table_id)


# ---------------------------------------------------------------- the lookup constraint system
class LookupCS:
    """LookupConstraintSystem::create without runtime tables (index.rs:188-430): selector columns, the concatenated
    fixed + gate tables padded with zeros to n - zk_rows - 1 rows, the table-id column (None if every id is 0)."""

    def __init__(self, p: int, gates: Sequence[str], fixed_tables, n: int, zk_rows: int, runtime_tables=None):
        """runtime_tables: None or [{"id", "first_column"}] (RuntimeTableCfg, runtime_tables.rs:20-50): their first columns are fixed, the
        second column is supplied per proof (index.rs:241-311)."""
        self.p, self.n, self.zk_rows = p, n, zk_rows
        self.info = LookupInfo(gates, uses_runtime_tables=runtime_tables is not None)
        assert self.info.patterns, "no lookup pattern in the circuit"
        max_entries = n - zk_rows - 1
        # selectors (lookups.rs:222-264)
        self.selectors = {q: [0] * n for q in self.info.patterns}
        gate_tables = set()
        for i, g in enumerate(gates[:n]):
            for r in (CURR, NEXT):
                pat = pattern_from_gate(g, r)
                if pat:
                    self.selectors[pat][i + r] = 1
                    if PATTERNS[pat]["table"]:
                        gate_tables.add(PATTERNS[pat]["table"])
        order = sorted(gate_tables, key=lambda t: 0 if t == "RangeCheck" else 1)       # Ord for GateLookupTable (mod.rs:26-37)
        tables = list(fixed_tables) + [range_check_table(p) if t == "RangeCheck" else xor_table(p) for t in order]
        self.runtime_tables = None if runtime_tables is None else [(rt["id"], len(rt["first_column"])) for rt in runtime_tables]
        self.runtime_offset, self.runtime_selector = None, None
        if runtime_tables is not None:
            assert len({rt["id"] for rt in runtime_tables}) == len(runtime_tables), "runtime table duplicates"
            self.runtime_offset = sum(len(t["data"][0]) for t in tables)
            rlen = sum(len(rt["first_column"]) for rt in runtime_tables)
            sel = [1] * self.runtime_offset + [0] * rlen + [1] * (n - self.runtime_offset - rlen)
            for r in range(n - zk_rows, n):
                sel[r] = 0
            self.runtime_selector = sel
            tables += [{"id": rt["id"], "data": [list(rt["first_column"]), [0] * len(rt["first_column"])]} for rt in runtime_tables]
        ids = [t["id"] for t in tables]
        assert len(set(ids)) == len(ids), "lookup table id collision"
        width = max([len(t["data"]) for t in tables] + [self.info.max_joint_size])
        cols = [[] for _ in range(width)]
        table_ids = []
        non_zero_id = False
        for t in tables:
            ln = len(t["data"][0])
            if t["id"] != 0:
                non_zero_id = True
            else:
                assert any(all(c[r] % p == 0 for c in t["data"]) for r in range(ln)), "table 0 needs a zero entry"
            table_ids += [t["id"] % p] * ln
            for k in range(width):
                cols[k] += [v % p for v in t["data"][k]] if k < len(t["data"]) else [0] * ln
        assert len(cols[0]) < max_entries, "lookup table too long for the domain"
        self.entries = len(cols[0])
        self.table_cols = [c + [0] * (n - len(c)) for c in cols]                 # d1 evaluations (rows above max_entries: 0)
        self.table_ids = (table_ids + [0] * (n - len(table_ids))) if non_zero_id else None
        self.dummy = ([], 0)                                                        # LookupConfiguration::new: all zeros in table 0

    def combiners(self, joint_combiner: int):
        tic = pow(joint_combiner, self.info.max_joint_size, self.p) if self.table_ids is not None else 0
        return joint_combiner % self.p, tic

    def constraint_combiners(self, joint_combiner: int):
        """The constraint EXPRESSIONS use joint_combiner^max_joint_size for the table id whether or not the index has a table-id
        column (constraints.rs:424-440); on the domain both agree (all table ids are 0 then), at zeta they do not."""
        return joint_combiner % self.p, pow(joint_combiner, self.info.max_joint_size, self.p)

    def joint_table(self, joint_combiner: int, runtime_second_column=None) -> List[int]:
        """The combined table on d1 (prover.rs:500-572, the stride-8 sub-grid of joint_lookup_table_d8); with runtime tables the second
        column is the fixed one + the proof's runtime contribution (prover.rs:455-464)."""
        jc, tic = self.combiners(joint_combiner)
        cols = self.table_cols
        if runtime_second_column is not None:
            cols = [c if k != 1 else [(a + b) % self.p for a, b in zip(c, runtime_second_column)] for k, c in enumerate(cols)]
        return [combine_table_entry(self.p, jc, tic, [c[r] for c in cols], self.table_ids[r] if self.table_ids else 0)
                for r in range(self.n)]

    def dummy_value(self, joint_combiner: int) -> int:
        jc, tic = self.combiners(joint_combiner)
        return combine_table_entry(self.p, jc, tic, self.dummy[0], self.dummy[1])


def zk_patch(vals: List[int], n: int, zk_rows: int, zk_values: Sequence[int]) -> List[int]:
    """constraints.rs:35-48 with the random tail passed in."""
    last = n - zk_rows
    assert len(vals) <= last and len(zk_values) == zk_rows
    return list(vals) + [0] * (last - len(vals)) + list(zk_values)


def sorted_columns(cs: LookupCS, gates: Sequence[str], witness, joint_combiner: int, table=None) -> List[List[int]]:
    """constraints.rs:90-194: the multiset table ++ lookups, sorted by the table and laid out as a snake over
    max_per_row + 1 columns of n - zk_rows values each (before zk_patch).  Raises ValueError(row) on a value
    that is not in the table (ProverError::ValueNotInTable)."""
    p, n = cs.p, cs.n
    jc, tic = cs.combiners(joint_combiner)
    table = table if table is not None else cs.joint_table(joint_combiner)
    dummy = cs.dummy_value(joint_combiner)
    lookup_rows = n - cs.zk_rows - 1
    mpr = cs.info.max_per_row
    counts: Dict[int, int] = {}
    for t in table[:lookup_rows]:
        counts.setdefault(t, 1)
    by_row = cs.info.by_row(gates)
    for i in range(lookup_rows):
        spec = by_row[i] if i < len(by_row) else []
        for jl in spec:
            v = spec_value(p, jl, jc, tic, witness, i)
            if v not in counts:
                raise ValueError(i)
            counts[v] += 1
        counts[dummy] = counts.get(dummy, 0) + (mpr - len(spec))
    cols = [[] for _ in range(mpr + 1)]
    i = 0
    for t in table[:lookup_rows]:
        c = counts[t]
        counts[t] = 1
        for j in range(c):
            cols[(i + j) // lookup_rows].append(t)
        i += c
    for k in range(mpr):
        cols[k].append(cols[k + 1][0])
    cols[mpr].append(cols[mpr][-1])
    for k in range(1, mpr + 1, 2):
        cols[k].reverse()
    return cols


def aggregation(cs: LookupCS, gates: Sequence[str], witness, joint_combiner: int, beta: int, gamma: int,
                sorted_cols: Sequence[Sequence[int]], zk_values: Optional[Sequence[int]], draw=None, table=None) -> List[int]:
    """constraints.rs:233-338; `sorted_cols` are the zk-patched columns (length n).  The random tail is `zk_values`, or drawn
    with `draw()` AFTER the running product, as the reference's zk_patch call does (constraints.rs:327)."""
    p, n = cs.p, cs.n
    jc, tic = cs.combiners(joint_combiner)
    table = table if table is not None else cs.joint_table(joint_combiner)
    dummy = cs.dummy_value(joint_combiner)
    lookup_rows = n - cs.zk_rows - 1
    mpr = cs.info.max_per_row
    beta1 = (1 + beta) % p
    gb1 = gamma * beta1 % p
    by_row = cs.info.by_row(gates)
    comp = [1]
    for _ in range(mpr):
        comp.append(comp[-1] * ((gamma + dummy) % p) % p)
    b1m = pow(beta1, mpr, p)
    comp = [c * b1m % p for c in comp]
    agg = [1]
    for row in range(lookup_rows):
        den = 1
        for k, s in enumerate(sorted_cols):
            i1, i2 = (row, row + 1) if k % 2 == 0 else (row + 1, row)
            den = den * ((gb1 + s[i1] + beta * s[i2]) % p) % p
        spec = by_row[row] if row < len(by_row) else []
        f = comp[mpr - len(spec)]
        for jl in spec:
            f = f * ((gamma + spec_value(p, jl, jc, tic, witness, row)) % p) % p
        t = (gb1 + table[row] + beta * table[row + 1]) % p
        agg.append(agg[-1] * f % p * t % p * pow(den, p - 2, p) % p)
    if zk_values is None:
        zk_values = [draw() for _ in range(cs.zk_rows)]
    return zk_patch(agg, n, cs.zk_rows, zk_values)


# ---------------------------------------------------------------- constraints, row by row
def unnormalized_lagrange_basis(p: int, omega: int, n: int, i: int, x: int) -> int:
    """expr.rs:140-147: (x^n - 1) / (x - omega^i); i may be negative."""
    wi = pow(omega, i % n, p)
    return (pow(x, n, p) - 1) * pow((x - wi) % p, p - 2, p) % p


def vanishes_on_last_n_rows(p: int, omega: int, n: int, i: int, x: int) -> int:
    """permutation.rs:78-89."""
    acc = 1
    for k in range(n - i, n):
        acc = acc * ((x - pow(omega, k, p)) % p) % p
    return acc


def constraint_values(cs: LookupCS, joint_combiner: int, beta: int, gamma: int, cell, atoms) -> List[int]:
    """The lookup constraints of constraints.rs:378-673 (generate_feature_flags = false) at ONE evaluation point, in the
    reference's order: aggregation equation, first / final value of the aggregation, max_per_row snake compatibility
    checks, zero padding up to 4.  `cell(kind, index, row)` returns a column value (kind in 'w', 'sorted', 'aggreg',
    'table', 'selector'); `atoms` = {'vanish': VanishesOnZeroKnowledgeAndPreviousRows, 'l0': UnnormalizedLagrangeBasis(0),
    'lfinal': UnnormalizedLagrangeBasis(-zk_rows - 1)} at that point."""
    p = cs.p
    info = cs.info
    jc, tic = cs.constraint_combiners(joint_combiner)
    mpr = info.max_per_row
    beta1 = (1 + beta) % p
    gb1 = gamma * beta1 % p
    dummy = cs.dummy_value(joint_combiner)
    b1m = pow(beta1, mpr, p)

    def f_term(spec):
        acc = pow((gamma + dummy) % p, mpr - len(spec), p) * b1m % p
        for tid, entries in spec:
            ev = lambda pos: cell("w", pos[1], pos[0])
            table_id = tid[1] % p if tid[0] == "const" else ev((CURR, tid[1]))
            vals = [sum(c * ev(pos) for c, pos in e) % p for e in entries]
            acc = acc * ((gamma + combine_table_entry(p, jc, tic, vals, table_id)) % p) % p
        return acc

    indicator = sum(cell("selector", q, CURR) for q in info.patterns) % p
    f_chunk = (1 - indicator) * f_term([]) % p
    for q in info.patterns:
        f_chunk = (f_chunk + cell("selector", q, CURR) * f_term(PATTERNS[q]["lookups"])) % p
    t_chunk = (gb1 + cell("table", 0, CURR) + beta * cell("table", 0, NEXT)) % p
    numerator = f_chunk * t_chunk % p
    denominator = 1
    for i in range(mpr + 1):
        s1, s2 = (CURR, NEXT) if i % 2 == 0 else (NEXT, CURR)
        denominator = denominator * ((gb1 + cell("sorted", i, s1) + beta * cell("sorted", i, s2)) % p) % p
    aggreg_eq = (cell("aggreg", 0, NEXT) * denominator - cell("aggreg", 0, CURR) * numerator) % p
    res = [atoms["vanish"] * aggreg_eq % p,
           atoms["l0"] * (cell("aggreg", 0, CURR) - 1) % p,
           atoms["lfinal"] * (cell("aggreg", 0, CURR) - 1) % p]
    for i in range(mpr):
        basis = atoms["lfinal"] if i % 2 == 0 else atoms["l0"]
        res.append(basis * (cell("sorted", i, CURR) - cell("sorted", i + 1, CURR)) % p)
    res += [0] * (4 - mpr)
    return res


def verify(cs: LookupCS, gates: Sequence[str], witness, joint_combiner: int, sorted_cols: Sequence[Sequence[int]]) -> None:
    """The reference's own checker of the sorted columns (constraints.rs:692-796): overlaps agree, the de-snaked
    sequence is sorted by the table, and it is multiset-equal to table ++ lookups (padded with dummies)."""
    p, n = cs.p, cs.n
    jc, tic = cs.combiners(joint_combiner)
    table = cs.joint_table(joint_combiner)
    dummy = cs.dummy_value(joint_combiner)
    lookup_rows = n - cs.zk_rows - 1
    for i in range(len(sorted_cols) - 1):
        pos = lookup_rows if i % 2 == 0 else 0
        assert sorted_cols[i][pos] == sorted_cols[i + 1][pos], ("overlap", i)
    joined = []
    for i, s in enumerate(sorted_cols):
        es = list(s[:lookup_rows + 1])
        joined += es if i % 2 == 0 else es[::-1]
    k = 0
    for t in table[:lookup_rows]:                      # (the reference's caller passes the number of table rows to walk; the padding rows hold the dummy)
        while k < len(joined) and joined[k] == t:
            k += 1
    assert k == len(joined), "not sorted by the table"
    sorted_counts: Dict[int, int] = {}
    for i, s in enumerate(sorted_cols):
        xs = s[:lookup_rows] if i % 2 == 0 else s[1:lookup_rows + 1]
        for x in xs:
            sorted_counts[x] = sorted_counts.get(x, 0) + 1
    all_lookups: Dict[int, int] = {}
    for t in table[:lookup_rows]:
        all_lookups[t] = all_lookups.get(t, 0) + 1
    by_row = cs.info.by_row(gates)
    for i in range(lookup_rows):
        spec = by_row[i] if i < len(by_row) else []
        for jl in spec:
            v = spec_value(p, jl, jc, tic, witness, i)
            all_lookups[v] = all_lookups.get(v, 0) + 1
        all_lookups[dummy] = all_lookups.get(dummy, 0) + cs.info.max_per_row - len(spec)
    assert all_lookups == sorted_counts, "multiset mismatch"
