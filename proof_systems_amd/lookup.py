"""The lookup argument of Kimchi's prover on the device, over the C ABI (no CPU fallback; host code only where the
reference's own code is a sequential host loop).

Reference (kimchi/src/): prover.rs:383-673 (the lookup part of ProverProof::create), circuits/lookup/index.rs:188-430
(LookupConstraintSystem::create), circuits/lookup/constraints.rs:90-194 (sorted), :233-338 (aggregation), :378-673
(constraints), circuits/lookup/lookups.rs:222-280 (selectors, by_row).

What runs where
  * `sorted`       the reference counts multiplicities in a HashMap and walks the table once (constraints.rs:108-171):
                   sequential host work, stays on the host here as well (numpy / dict over the values' limbs).
  * `aggregation`  per-row numerators f_chunk * t_chunk and denominators s_chunk: two token programs run by
                   kh_expr_evaluations_dev over the device-resident witness / table / sorted columns, one
                   kh_batch_inversion_dev, one running product kh_field_scan_dev -- the same three vector steps as the
                   permutation accumulator (permutation.rs:447-575).
  * constraints    one token program (proof_systems_amd.polish.lookup_program) evaluated on d8 next to the gates.
The row-set atoms of the constraints (VanishesOnZeroKnowledgeAndPreviousRows, UnnormalizedLagrangeBasis(0 / -zk_rows - 1),
expr.rs:883-893) are provided as d8 evaluation columns computed once per index (`atom_columns`).
"""
from typing import Dict, List, Sequence

import numpy as np

from . import khip
from . import polish as OP
from .prover import Fld

CURR, NEXT = 0, 1


def pattern_from_gate(typ: str, row: int):
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


def gate_table(name: str):
    """tables/xor.rs:9-30 (reversed: the last row is (0, 0, 0)) and tables/range_check.rs:10-22."""
    if name == "Xor":
        rows = [(i, j, i ^ j) for i in range(16) for j in range(16)][::-1]
        return {"id": OP.LOOKUP_XOR_TABLE_ID, "data": [[r[k] for r in rows] for k in range(3)]}
    return {"id": OP.LOOKUP_RANGE_CHECK_TABLE_ID, "data": [list(range(1 << 12))]}


class LookupIndex:
    """LookupConstraintSystem::create without runtime tables: the selector columns, the concatenated table columns and
    the table-id column as d1 evaluations on the device (+ host copies as integers for `sorted`)."""

    def __init__(self, fid: int, gates: Sequence[str], fixed_tables, log2_n: int, zk_rows: int = 3, runtime_tables=None):
        """runtime_tables: None or [{"id", "first_column"}] (RuntimeTableCfg): tables whose first column is fixed in the index and whose
        second column arrives with each proof (index.rs:241-311)."""
        self.fid, self.F = fid, Fld(fid)
        self.logn, self.n, self.zk_rows = log2_n, 1 << log2_n, zk_rows
        n, p = self.n, self.F.p
        used, sel, gate_tables = set(), {}, set()
        self.row_pattern: List = [None] * (n + 1)
        for i, g in enumerate(gates[:n]):
            for r in (CURR, NEXT):
                pat = pattern_from_gate(g, r)
                if pat:
                    used.add(pat)
                    sel.setdefault(pat, np.zeros(n, dtype=np.int64))[i + r] = 1
                    self.row_pattern[i + r] = pat
                    if pat in ("Xor", "RangeCheck", "ForeignFieldMul"):
                        gate_tables.add("Xor" if pat == "Xor" else "RangeCheck")
        if not used:
            raise ValueError("no lookup pattern in the circuit")
        self.patterns = [q for q in OP.LOOKUP_PATTERN_ORDER if q in used]
        self.max_per_row = OP.lookup_max_per_row(self.patterns)
        self.max_joint_size = OP.lookup_max_joint_size(self.patterns)
        self.joint_lookup_used = self.max_joint_size > 1
        tables = list(fixed_tables) + [gate_table(t) for t in sorted(gate_tables, key=lambda t: 0 if t == "RangeCheck" else 1)]
        self.runtime_tables = None if runtime_tables is None else [(rt["id"], len(rt["first_column"])) for rt in runtime_tables]
        self.runtime_offset, self.runtime_selector = None, None
        if runtime_tables is not None:
            if len({rt["id"] for rt in runtime_tables}) != len(runtime_tables):
                raise ValueError("runtime table duplicates")
            self.runtime_offset = sum(len(t["data"][0]) for t in tables)
            rlen = sum(len(rt["first_column"]) for rt in runtime_tables)
            rsel = [1] * self.runtime_offset + [0] * rlen + [1] * (n - self.runtime_offset - rlen)
            rsel[n - zk_rows:] = [0] * zk_rows
            self.runtime_selector = rsel
            tables += [{"id": rt["id"], "data": [list(rt["first_column"]), [0] * len(rt["first_column"])]} for rt in runtime_tables]
        ids = [t["id"] for t in tables]
        if len(set(ids)) != len(ids):
            raise ValueError("lookup table id collision")
        width = max([len(t["data"]) for t in tables] + [self.max_joint_size])
        cols: List[List[int]] = [[] for _ in range(width)]
        tids: List[int] = []
        for t in tables:
            ln = len(t["data"][0])
            if t["id"] == 0 and not any(all(c[r] % p == 0 for c in t["data"]) for r in range(ln)):
                raise ValueError("table 0 needs a zero entry")
            tids += [t["id"] % p] * ln
            for k in range(width):
                cols[k] += [v % p for v in t["data"][k]] if k < len(t["data"]) else [0] * ln
        if len(cols[0]) >= n - zk_rows - 1:
            raise ValueError("lookup table too long for the domain")
        self.entries = len(cols[0])
        self.table_cols = [c + [0] * (n - len(c)) for c in cols]
        self.table_ids = (tids + [0] * (n - len(tids))) if any(i != 0 for i in ids) else None
        self.selectors = {q: [int(v) for v in sel[q]] for q in self.patterns}
        up = lambda vals: khip.DevBuf(n * 32).upload(self.F.limbs_many(vals))
        self.d_selectors = {q: up(self.selectors[q]) for q in self.patterns}
        self.d_table_cols = [up(c) for c in self.table_cols]
        self.d_table_ids = up(self.table_ids) if self.table_ids is not None else None
        self.d_runtime_selector = up(self.runtime_selector) if self.runtime_selector is not None else None

    def combiners(self, joint_combiner: int):
        p = self.F.p
        return joint_combiner % p, (pow(joint_combiner, self.max_joint_size, p) if self.table_ids is not None else 0)

    def constraint_combiners(self, joint_combiner: int):
        """For the constraint token program: the expressions use joint_combiner^max_joint_size for the table id whether or not the
        index has a table-id column (constraints.rs:424-440) -- equal on the domain, different at zeta, and the verifier's."""
        return joint_combiner % self.F.p, pow(joint_combiner, self.max_joint_size, self.F.p)

    # ---- the combined table (prover.rs:500-572) on the device: Horner over the table columns + table_id_combiner * ids
    def joint_table_dev(self, joint_combiner: int, runtime_dev=None) -> "khip.DevBuf":
        """runtime_dev: the proof's runtime contribution on d1 (added to the second table column, prover.rs:455-464)."""
        jc, tic = self.combiners(joint_combiner)
        ncol = len(self.d_table_cols)
        nb = ncol + (1 if self.d_table_ids is not None else 0)          # buffer index of the runtime column
        col = lambda k: [OP.cell(k)] + ([OP.cell(nb), (OP.TOK_ADD, 0)] if (k == 1 and runtime_dev is not None) else [])
        toks = col(ncol - 1)
        for k in range(ncol - 2, -1, -1):
            toks += [(OP.TOK_CONST, 0), (OP.TOK_MUL, 0)] + col(k) + [(OP.TOK_ADD, 0)]
        bufs = list(self.d_table_cols)
        if self.d_table_ids is not None:
            toks += [(OP.TOK_CONST, 1), OP.cell(ncol), (OP.TOK_MUL, 0), (OP.TOK_ADD, 0)]
            bufs.append(self.d_table_ids)
        if runtime_dev is not None:
            bufs.append(runtime_dev)
        out = khip.DevBuf(self.n * 32)
        khip.expr_evaluations_dev(self.fid, toks, bufs, [self.n] * len(bufs), self.F.limbs_many([jc, tic]), self.n, out, stride=1, next_shift=1)
        return out

    def free(self):
        for b in list(self.d_selectors.values()) + self.d_table_cols + ([self.d_table_ids] if self.d_table_ids is not None else []) + \
                ([self.d_runtime_selector] if self.d_runtime_selector is not None else []):
            b.free()


def _joint_value_host(F: Fld, jc: int, tic: int, vals, table_id: int) -> int:
    acc = 0
    for x in reversed(vals):
        acc = (jc * acc + x) % F.p
    return (acc + tic * table_id) % F.p


def sorted_columns(ix: LookupIndex, witness: Sequence[Sequence[int]], joint_table: Sequence[int], joint_combiner: int) -> List[List[int]]:
    """constraints.rs:90-194 (host, as in the reference): `witness[col][row]` and the combined table as integers.  Returns the
    max_per_row + 1 snake columns of n - zk_rows values each; ValueError(row) for a value that is not in the table."""
    F, n = ix.F, ix.n
    jc, tic = ix.combiners(joint_combiner)
    lookup_rows = n - ix.zk_rows - 1
    mpr = ix.max_per_row
    dummy = 0                                                   # LookupConfiguration::new: the all-zero entry of table 0
    counts: Dict[int, int] = {}
    for t in joint_table[:lookup_rows]:
        counts.setdefault(t, 1)
    for i in range(lookup_rows):
        pat = ix.row_pattern[i]
        spec = OP.LOOKUP_PATTERNS[pat] if pat else []
        for tid, entry in spec:
            table_id = witness[tid[1]][i] if isinstance(tid, tuple) else tid
            v = _joint_value_host(F, jc, tic, [witness[c][i] for c in entry], table_id)
            if v not in counts:
                raise ValueError(i)
            counts[v] += 1
        counts[dummy] = counts.get(dummy, 0) + (mpr - len(spec))
    cols: List[List[int]] = [[] for _ in range(mpr + 1)]
    i = 0
    for t in joint_table[:lookup_rows]:
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


def lookup_values_dev(ix: LookupIndex, d_witness, joint_combiner: int) -> "khip.DevBuf":
    """The looked-up joint values on the device: max_per_row columns of n values, slot s of row r = sum_k jc^k w[entry_k][r] + tic * table_id for
    the s-th lookup of the row's pattern (lookups.rs:417-487), 0 -- the dummy entry's value -- where the row looks up fewer than max_per_row."""
    F, n = ix.F, ix.n
    jc, tic = ix.combiners(joint_combiner)
    out = khip.DevBuf(ix.max_per_row * n * 32)
    bufs = list(d_witness) + [ix.d_selectors[q] for q in ix.patterns]
    for s in range(ix.max_per_row):
        consts, toks, first = [jc, tic], [], True
        for k, q in enumerate(ix.patterns):
            spec = OP.LOOKUP_PATTERNS[q]
            if s >= len(spec):
                continue
            tid, entry = spec[s]
            t = [OP.cell(entry[-1])]
            for c in reversed(entry[:-1]):
                t += [(OP.TOK_CONST, 0), (OP.TOK_MUL, 0), OP.cell(c), (OP.TOK_ADD, 0)]
            if isinstance(tid, tuple):
                t += [OP.cell(tid[1]), (OP.TOK_CONST, 1), (OP.TOK_MUL, 0), (OP.TOK_ADD, 0)]
            elif tid:
                consts.append(tic * tid % F.p)
                t += [(OP.TOK_CONST, len(consts) - 1), (OP.TOK_ADD, 0)]
            toks += t + [OP.cell(15 + k), (OP.TOK_MUL, 0)] + ([] if first else [(OP.TOK_ADD, 0)])
            first = False
        khip.expr_evaluations_dev(ix.fid, toks, bufs, [n] * len(bufs), F.limbs_many(consts), n, out, stride=1, next_shift=1, out_offset=s * n)
    return out


def sorted_columns_dev(ix: LookupIndex, d_witness, d_table, joint_combiner: int):
    """`sorted` (constraints.rs:90-194) without Python integers: looked-up values by the device, the hash join natively on the host
    (kh_lookup_sorted).  Returns (max_per_row + 1, n - zk_rows, 4) limbs; ValueError(row) for a value that is not in the table."""
    n = ix.n
    d_vals = lookup_values_dev(ix, d_witness, joint_combiner)
    vals = d_vals.download((ix.max_per_row, n, 4))
    d_vals.free()
    return khip.lookup_sorted(d_table.download((n, 4)), n - ix.zk_rows - 1, vals, ix.max_per_row)


def zk_patch(F: Fld, vals: Sequence[int], n: int, zk_rows: int, rng) -> List[int]:
    """constraints.rs:35-48."""
    return list(vals) + [0] * (n - zk_rows - len(vals)) + [F.rand(rng) for _ in range(zk_rows)]


def column_layout(ix: LookupIndex, w0: int = 0):
    """Column numbers of the token programs: witness w0 .. w0 + 14, then sorted (max_per_row + 1), aggreg, table,
    one selector per pattern, and the three atoms."""
    c = w0 + 15
    cols = {"sorted": list(range(c, c + ix.max_per_row + 1))}
    c += ix.max_per_row + 1
    cols["aggreg"], cols["table"] = c, c + 1
    c += 2
    cols["selector"] = {q: c + k for k, q in enumerate(ix.patterns)}
    c += len(ix.patterns)
    cols["vanish"], cols["l0"], cols["lfinal"] = c, c + 1, c + 2
    cols["count"] = c + 3
    if ix.runtime_selector is not None:
        cols["runtime"], cols["runtime_selector"] = c + 3, c + 4
        cols["count"] = c + 5
    return cols


def aggregation_dev(ix: LookupIndex, d_witness, d_sorted, d_table, joint_combiner: int, beta: int, gamma: int, rng) -> "khip.DevBuf":
    """constraints.rs:233-338 on the device: aggreg[0] = 1, aggreg[i + 1] = aggreg[i] * f_chunk_i * t_chunk_i / s_chunk_i for the
    n - zk_rows - 1 lookup rows, random values in the zk rows.  All inputs are d1 columns on the device (DevBuf)."""
    F, n, fid = ix.F, ix.n, ix.fid
    jc, tic = ix.combiners(joint_combiner)
    cols = column_layout(ix)
    lookup_rows = n - ix.zk_rows - 1
    (num_t, num_c), (den_t, den_c) = OP.lookup_aggregation_programs(F.p, ix.patterns, cols, jc, tic, beta, gamma)
    dummy = khip.DevBuf(32)                                      # columns the aggregation programs never touch
    bufs = list(d_witness) + list(d_sorted) + [dummy, d_table] + [ix.d_selectors[q] for q in ix.patterns] + [dummy] * (cols["count"] - cols["vanish"])
    lens = [n] * 15 + [n] * len(d_sorted) + [1, n] + [n] * len(ix.patterns) + [1] * (cols["count"] - cols["vanish"])
    num, den, agg = khip.DevBuf(n * 32), khip.DevBuf(n * 32), khip.DevBuf(n * 32)
    num.zero(); den.zero()
    khip.expr_evaluations_dev(fid, num_t, bufs, lens, F.limbs_many(num_c), lookup_rows, num, stride=1, next_shift=1, out_offset=1)
    khip.expr_evaluations_dev(fid, den_t, bufs, lens, F.limbs_many(den_c), lookup_rows, den, stride=1, next_shift=1, out_offset=1)
    khip.batch_inversion_dev(fid, den, lookup_rows, offset=1)
    num.upload_at(0, F.limbs(1).reshape(1, 4)); den.upload_at(0, F.limbs(1).reshape(1, 4))
    khip.expr_evaluations_dev(fid, [OP.cell(0), OP.cell(1), (OP.TOK_MUL, 0)], [num, den], [n, n], F.limbs(1).reshape(1, 4), n, agg, stride=1, next_shift=1)
    khip.field_scan_dev(fid, khip.SCAN_MUL, agg, lookup_rows + 1)
    agg.upload_at((n - ix.zk_rows) * 32, F.limbs_many([F.rand(rng) for _ in range(ix.zk_rows)]))
    for b in (num, den, dummy):
        b.free()
    return agg


def atom_columns(ix: LookupIndex, log2_blowup: int = 3):
    """d8 evaluations of the three row-set atoms (expr.rs:883-893), computed on the host once per index:
    vanish = prod_{k = n - zk_rows - 1}^{n - 1} (x - w^k), l0 = (x^n - 1) / (x - 1), lfinal = (x^n - 1) / (x - w^(n - zk_rows - 1))."""
    F, n, p = ix.F, ix.n, ix.F.p
    m = n << log2_blowup
    root = khip_root(F, ix.logn + log2_blowup)
    w = pow(root, 1 << log2_blowup, p)
    xs = [1] * m
    for k in range(1, m):
        xs[k] = xs[k - 1] * root % p
    last = [pow(w, k, p) for k in range(n - ix.zk_rows - 1, n)]
    wf = pow(w, n - ix.zk_rows - 1, p)
    vanish, l0, lfinal = [], [], []
    for k, x in enumerate(xs):
        v = 1
        for t in last:
            v = v * (x - t) % p
        vanish.append(v)
        zh = (pow(x, n, p) - 1) % p
        if k % (1 << log2_blowup) == 0:                            # a point of d1: the quotient's limit n * w^(-i) at x = w^i, else 0
            r = k >> log2_blowup
            l0.append(n % p if r == 0 else 0)
            lfinal.append(n * pow(wf, p - 2, p) % p if r == n - ix.zk_rows - 1 else 0)
        else:
            l0.append(zh * pow((x - 1) % p, p - 2, p) % p)
            lfinal.append(zh * pow((x - wf) % p, p - 2, p) % p)
    return [khip.DevBuf(m * 32).upload(F.limbs_many(c)) for c in (vanish, l0, lfinal)]


def atom_columns_dev(ix: LookupIndex, x8, log2_blowup: int = 3):
    """The same three columns computed on the device from the d8 evaluations of x (`x8`: DevBuf / view of 8n elements): products of (x - w^k) for
    `vanish`; (x^n - 1) * (x - a)^-1 with one batched inversion per atom, the removable singularity (x = a, a point of d1) patched with its limit
    n * a^-1.  atom_columns above is the host restatement this is tested against (tests/test_gpu_lookup.py); at 2^16 rows it takes ~50 s, this ~1 ms."""
    F, n, p, fid = ix.F, ix.n, ix.F.p, ix.fid
    m = n << log2_blowup
    w = khip_root(F, ix.logn)
    last = [pow(w, k, p) for k in range(n - ix.zk_rows - 1, n)]
    wf = pow(w, n - ix.zk_rows - 1, p)
    out = []
    toks = []
    for i in range(len(last)):
        toks += [OP.cell(0), (OP.TOK_CONST, i), (OP.TOK_SUB, 0)] + ([(OP.TOK_MUL, 0)] if i else [])
    vanish = khip.DevBuf(m * 32)
    khip.expr_evaluations_dev(fid, toks, [x8], [m], F.limbs_many(last), m, vanish, stride=1, next_shift=1)
    out.append(vanish)
    for a, row in ((1, 0), (wf, (n - ix.zk_rows - 1) << log2_blowup)):
        den = khip.DevBuf(m * 32)
        khip.expr_evaluations_dev(fid, [OP.cell(0), (OP.TOK_CONST, 0), (OP.TOK_SUB, 0)], [x8], [m], F.limbs_many([a]), m, den, stride=1, next_shift=1)
        den.upload_at(row * 32, F.limbs(1).reshape(1, 4))                # x = a: any non-zero value, the product below is 0 there and patched
        khip.batch_inversion_dev(fid, den, m)
        col = khip.DevBuf(m * 32)
        khip.expr_evaluations_dev(fid, [OP.cell(0), (OP.TOK_POW, n), (OP.TOK_CONST, 0), (OP.TOK_SUB, 0), OP.cell(1), (OP.TOK_MUL, 0)], [x8, den], [m, m],
                                  F.limbs_many([1]), m, col, stride=1, next_shift=1)
        col.upload_at(row * 32, F.limbs(n * pow(a, p - 2, p) % p).reshape(1, 4))
        khip.sync()
        den.free()
        out.append(col)
    return out


def khip_root(F: Fld, log2_n: int) -> int:
    """w_{2^k} = (5^T)^(2^(32 - k)), T = (p - 1) >> 32 (kimchi/src/circuits/domains.rs:40-69)."""
    return pow(pow(5, (F.p - 1) >> 32, F.p), 1 << (32 - log2_n), F.p)
