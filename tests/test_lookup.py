"""The oracle's restatement of the lookup argument (oracle/lookup.py) on a small circuit with two lookup patterns (Xor16 rows
into the 4-bit XOR table, Lookup rows into a user table with id 2): the sorted columns pass the reference's own checker
(`lookup::constraints::verify`, constraints.rs:692-796), the aggregation ends at 1 (constraints.rs:325-331), every constraint
of constraints.rs:378-673 vanishes on every row of a satisfied witness, and each of them reacts to tampering.  No GPU."""
import random

import pytest

from oracle import lookup as L
from oracle import pasta as P

F = P.Fp
p = F.p
LOGN, ZK = 9, 3
N = 1 << LOGN


def user_table():
    return {"id": 2, "data": [[i for i in range(16)], [(7 * i * i + 3) % 1000 for i in range(16)]]}


def circuit(rnd, nxor=40, nlookup=30):
    """gate names per row and a witness whose lookups are all in the tables (other cells random)."""
    gates = ["Generic"] * (N - ZK)
    rows = rnd.sample(range(0, N - ZK - 2), nxor + nlookup)
    wit = [[rnd.randrange(p) for _ in range(N)] for _ in range(15)]
    tab = user_table()
    for k, r in enumerate(rows):
        if k < nxor:
            gates[r] = "Xor16"
            for i in range(4):
                a, b = rnd.randrange(16), rnd.randrange(16)
                wit[3 + i][r], wit[7 + i][r], wit[11 + i][r] = a, b, a ^ b
        else:
            gates[r] = "Lookup"
            wit[0][r] = 2
            for i in range(3):
                e = rnd.randrange(16)
                wit[2 * i + 1][r], wit[2 * i + 2][r] = tab["data"][0][e], tab["data"][1][e]
    return gates, wit


def setup(seed=5):
    rnd = random.Random(seed)
    gates, wit = circuit(rnd)
    cs = L.LookupCS(p, gates, [user_table()], N, ZK)
    jc, beta, gamma = rnd.randrange(p), rnd.randrange(p), rnd.randrange(p)
    sorted_cols = L.sorted_columns(cs, gates, wit, jc)
    sorted_cols = [L.zk_patch(c, N, ZK, [rnd.randrange(p) for _ in range(ZK)]) for c in sorted_cols]
    agg = L.aggregation(cs, gates, wit, jc, beta, gamma, sorted_cols, [rnd.randrange(p) for _ in range(ZK)])
    return rnd, gates, wit, cs, jc, beta, gamma, sorted_cols, agg


def rows_of_constraints(cs, wit, jc, beta, gamma, sorted_cols, agg):
    """constraint values on every point of d1 (x = omega^r)."""
    omega = F.root_of_unity(LOGN)
    table = cs.joint_table(jc)
    out = []
    for r in range(N):
        x = pow(omega, r, p)
        cols = {"w": wit, "sorted": sorted_cols, "aggreg": [agg], "table": [table]}

        def cell(kind, idx, row, r=r):
            if kind == "selector":
                return cs.selectors[idx][(r + row) % N]
            return cols[kind][idx][(r + row) % N]
        # on the domain the Lagrange atoms are 0 / non-zero indicators; use their exact values via the limit form
        atoms = {"vanish": L.vanishes_on_last_n_rows(p, omega, N, ZK + 1, x),
                 "l0": N % p if r == 0 else 0,                                        # (x^n - 1)/(x - 1) at x = 1 is n
                 "lfinal": (N * pow(pow(omega, (-(ZK + 1)) % N, p), N - 1, p)) % p if r == N - ZK - 1 else 0}
        out.append(L.constraint_values(cs, jc, beta, gamma, cell, atoms))
    return out


def test_info_and_tables():
    rnd, gates, wit, cs, *_ = setup()
    assert cs.info.patterns == ["Xor", "Lookup"]
    assert (cs.info.max_per_row, cs.info.max_joint_size, cs.info.joint_lookup_used) == (4, 3, True)
    assert cs.entries == 16 + 256 and cs.table_ids is not None and len(cs.table_cols) == 3
    assert cs.table_cols[0][16 + 255] == 0 and cs.table_cols[2][16] == (15 ^ 15)     # the XOR table is reversed: last row (0, 0, 0)
    assert L.combine_table_entry(p, 3, 5, [1, 2, 4], 7) == (1 + 3 * 2 + 9 * 4 + 5 * 7) % p


def test_sorted_passes_the_reference_checker_and_aggregation_ends_at_one():
    rnd, gates, wit, cs, jc, beta, gamma, sorted_cols, agg = setup()
    assert len(sorted_cols) == cs.info.max_per_row + 1 and all(len(c) == N for c in sorted_cols)
    L.verify(cs, gates, wit, jc, sorted_cols)
    assert agg[0] == 1 and agg[N - ZK - 1] == 1
    bad = [list(c) for c in sorted_cols]
    bad[1][7], bad[1][8] = bad[1][8], (bad[1][7] + 1) % p
    with pytest.raises(AssertionError):
        L.verify(cs, gates, wit, jc, bad)


def test_value_outside_the_table_is_reported():
    rnd, gates, wit, cs, jc, *_ = setup()
    r = gates.index("Xor16")
    wit[11][r] = (wit[11][r] + 1) % 16 if (wit[3][r] ^ wit[7][r]) != (wit[11][r] + 1) % 16 else (wit[11][r] + 2) % 16
    with pytest.raises(ValueError) as e:
        L.sorted_columns(cs, gates, wit, jc)
    assert e.value.args[0] == r


def test_constraints_vanish_on_a_satisfied_witness_and_react_to_tampering():
    rnd, gates, wit, cs, jc, beta, gamma, sorted_cols, agg = setup()
    rows = rows_of_constraints(cs, wit, jc, beta, gamma, sorted_cols, agg)
    assert len(rows[0]) == 3 + 4
    assert all(v == 0 for row in rows for v in row)
    # tamper with one aggregation value: the aggregation equation breaks on the two rows that see it
    agg2 = list(agg); agg2[100] = (agg2[100] + 1) % p
    rows = rows_of_constraints(cs, wit, jc, beta, gamma, sorted_cols, agg2)
    assert [r for r in range(N) if rows[r][0]] == [99, 100]
    # a witness cell that is looked up
    r = gates.index("Lookup")
    wit2 = [list(c) for c in wit]; wit2[2][r] = (wit2[2][r] + 1) % p
    rows = rows_of_constraints(cs, wit2, jc, beta, gamma, sorted_cols, agg)
    assert [q for q in range(N) if rows[q][0]] == [r]
    # first / final values and the snake turns
    agg3 = list(agg); agg3[0] = 2
    assert rows_of_constraints(cs, wit, jc, beta, gamma, sorted_cols, agg3)[0][1] != 0
    s2 = [list(c) for c in sorted_cols]; s2[1][0] = (s2[1][0] + 1) % p                 # column 1 meets column 2 at row 0
    rows = rows_of_constraints(cs, wit, jc, beta, gamma, s2, agg)
    assert rows[0][3 + 1] != 0
    s3 = [list(c) for c in sorted_cols]; s3[0][N - ZK - 1] = (s3[0][N - ZK - 1] + 1) % p  # column 0 meets column 1 at the final lookup row
    rows = rows_of_constraints(cs, wit, jc, beta, gamma, s3, agg)
    assert rows[N - ZK - 1][3 + 0] != 0
