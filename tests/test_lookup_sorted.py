"""kh_lookup_sorted (csrc/host_lookup.cpp) -- the `sorted` step of the lookup argument, kimchi/src/circuits/lookup/constraints.rs:90-194, as native
host code -- against the oracle's restatement of the same function (oracle/lookup.py::sorted_columns' counting and snake layout, here on opaque
32-byte values): tables with repeated entries, the dummy value, 1 to 4 lookups per row, a looked-up value that is not in the table.  No GPU."""
import random

import numpy as np
import pytest


def reference_sorted(table, values, lookup_rows, mpr):
    """the reference's algorithm on hashable values: counts over the table (a repeated entry counts once), every entry repeated count times in
    table order, cut into mpr + 1 columns sharing one element, odd columns reversed"""
    counts = {}
    for t in table[:lookup_rows]:
        counts.setdefault(t, 1)
    for r in range(lookup_rows):
        for s in range(mpr):
            v = values[s][r]
            if v not in counts:
                raise ValueError(r)
            counts[v] += 1
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


@pytest.mark.parametrize("lookup_rows,distinct,mpr", [(1, 1, 1), (12, 5, 3), (60, 60, 4), (500, 37, 3), (4092, 1500, 4), (300, 299, 1)])
def test_sorted_columns_equal_the_reference_algorithm(lookup_rows, distinct, mpr):
    import proof_systems_amd.khip as khip
    rnd = random.Random(1000 * lookup_rows + mpr)
    n = lookup_rows + 4                                         # the arrays are longer than the rows used (zero-knowledge rows behind them)
    pool = [(0, 0, 0, 0)] + [tuple(rnd.getrandbits(64) for _ in range(4)) for _ in range(distinct - 1)]
    table = [pool[k] if k < distinct else pool[rnd.randrange(distinct)] for k in range(lookup_rows)] + [tuple(rnd.getrandbits(64) for _ in range(4)) for _ in range(4)]
    values = [[pool[rnd.randrange(distinct)] if rnd.random() < 0.8 else (0, 0, 0, 0) for _ in range(n)] for _ in range(mpr)]
    want = reference_sorted(table, values, lookup_rows, mpr)
    got = khip.lookup_sorted(np.array(table, dtype=np.uint64), lookup_rows, np.array(values, dtype=np.uint64), mpr)
    assert got.shape == (mpr + 1, lookup_rows + 1, 4)
    assert [[tuple(int(x) for x in v) for v in col] for col in got] == want
    # a value outside the table: the row comes back, as ProverError::ValueNotInTable(row)
    bad_row = rnd.randrange(lookup_rows)
    values[mpr - 1][bad_row] = (1, 2, 3, 4)
    with pytest.raises(ValueError) as e:
        khip.lookup_sorted(np.array(table, dtype=np.uint64), lookup_rows, np.array(values, dtype=np.uint64), mpr)
    assert e.value.args[0] == bad_row
    # ... but only rows below lookup_rows are looked at
    values[mpr - 1][bad_row] = (0, 0, 0, 0)
    values[0][lookup_rows] = (9, 9, 9, 9)
    khip.lookup_sorted(np.array(table, dtype=np.uint64), lookup_rows, np.array(values, dtype=np.uint64), mpr)
