"""o1vm's second prover (o1vm/src/pickles/prover.rs) as another caller of the hot path: the device prover of proof_systems_amd/o1vm.py
against the oracle's restatement (oracle/o1vm.py) -- same commitments, evaluations and opening for the same blinders, and the oracle's
restatement of pickles/verifier.rs accepts the device proof.  The circuit of the reference's own test (pickles/tests.rs:35-95: domain 8,
Pallas, the sum of all relation columns), and a 2^10 trace with products, inverses, a next-row access and the 72 dynamic selectors."""
import random

import numpy as np
import pytest

from oracle import cref
from oracle import o1vm as O
from oracle import pasta as P
from oracle import prover as OPR
from oracle import views as V

pytestmark = pytest.mark.gpu
TC, TK, TA, TM, TS = P.TOK_CONST, P.TOK_CELL, P.TOK_ADD, P.TOK_MUL, P.TOK_SUB


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def small_circuit(F, logn=3):
    """pickles/tests.rs:35-70"""
    p, n = F.p, 1 << logn
    z = list(range(n))
    inp = {"scratch": [list(z) for _ in range(O.SCRATCH_SIZE)], "scratch_inverse": [[0] * n for _ in range(O.SCRATCH_SIZE_INVERSE)], "lookup_state": [],
           "instruction_counter": [i + 1 for i in range(n)], "error": [(-(i * O.SCRATCH_SIZE + (i + 1))) % p for i in range(n)], "selector": list(z)}
    ncol = O.SCRATCH_SIZE + O.SCRATCH_SIZE_INVERSE + 2
    toks = [(TK, 0)]
    for c in range(1, ncol):
        toks += [(TK, 2 * c), (TA, 0)]
    return inp, [(toks, [])]


def trace_circuit(F, logn=10, seed=3):
    """a synthetic trace: scratch_0 * (1 / scratch_0) = 1; selector_j * (scratch_1 - j) = 0 for every instruction j; two lookup-state columns
    with ls_1 = ls_0^2; the instruction counter steps by one (next-row access; the wrap-around row is excused through the error column)."""
    p, n = F.p, 1 << logn
    rnd = random.Random(seed)
    s = [[rnd.randrange(p) for _ in range(n)] for _ in range(O.SCRATCH_SIZE)]
    s[0] = [rnd.randrange(1, p) for _ in range(n)]
    s[1] = [rnd.randrange(O.N_MIPS_SEL_COLS) for _ in range(n)]
    sinv = [[rnd.randrange(p) for _ in range(n)] for _ in range(O.SCRATCH_SIZE_INVERSE)]
    sinv[0] = list(s[0])
    ls0 = [rnd.randrange(p) for _ in range(n)]
    ic = [5 + i for i in range(n)]
    err = [0] * (n - 1) + [(ic[0] - ic[n - 1] - 1) % p]
    inp = {"scratch": s, "scratch_inverse": sinv, "lookup_state": [ls0, [x * x % p for x in ls0]], "instruction_counter": ic, "error": err, "selector": list(s[1])}
    L = 2
    c_ic, c_err, c_sel0 = O.SCRATCH_SIZE + O.SCRATCH_SIZE_INVERSE + L, O.SCRATCH_SIZE + O.SCRATCH_SIZE_INVERSE + L + 1, O.SCRATCH_SIZE + O.SCRATCH_SIZE_INVERSE + L + 2
    cons = [([(TK, 0), (TK, 2 * O.SCRATCH_SIZE), (TM, 0), (TC, 0), (TS, 0)], [1])]
    cons += [([(TK, 2 * (c_sel0 + j)), (TK, 2), (TC, 0), (TS, 0), (TM, 0)], [j]) for j in range(O.N_MIPS_SEL_COLS)]
    c_ls = O.SCRATCH_SIZE + O.SCRATCH_SIZE_INVERSE
    cons.append(([(TK, 2 * c_ls), (TK, 2 * c_ls), (TM, 0), (TK, 2 * (c_ls + 1)), (TS, 0)], []))
    cons.append(([(TK, 2 * c_ic + 1), (TK, 2 * c_ic), (TS, 0), (TC, 0), (TS, 0), (TK, 2 * c_err), (TS, 0)], [1]))
    return inp, cons


@pytest.mark.parametrize("cid,logn,which", [(1, 3, "small"), (0, 10, "trace"), (1, 10, "trace")])
def test_device_o1vm_proof_equals_the_oracle_and_verifies(khip, cid, logn, which):
    from proof_systems_amd import o1vm
    C = P.CURVES[cid]; F = C.scalar
    inp, cons = small_circuit(F, logn) if which == "small" else trace_circuit(F, logn)
    n = 1 << logn
    seed = bytes([90 + cid] * 32)
    osrs = OPR.Srs(C, n)
    want = O.prove(C, logn, osrs, inp, cons, P.StdRng(seed))
    assert O.verify(C, logn, osrs, cons, want, P.StdRng(bytes([2] * 32)))
    lim = lambda col: cref.ints_to_limbs([F.to_mont(v % F.p) for v in col])
    dinp = {"scratch": np.stack([lim(c) for c in inp["scratch"]]), "scratch_inverse": np.stack([lim(c) for c in inp["scratch_inverse"]]),
            "lookup_state": np.stack([lim(c) for c in inp["lookup_state"]]) if inp["lookup_state"] else np.zeros((0, n, 4), np.uint64),
            "instruction_counter": lim(inp["instruction_counter"]), "error": lim(inp["error"]), "selector": inp["selector"]}
    srs = khip.Srs.create(cid, n)
    got = o1vm.prove(cid, logn, srs, dinp, cons, V.RefRng(P.StdRng(seed)))
    ch = lambda t: V.chunks(C, t)
    pr = {"commitments": [ch(t) for t in got["commitments"]], "zeta_evaluations": got["zeta_evaluations"], "zeta_omega_evaluations": got["zeta_omega_evaluations"],
          "quotient_commitment": ch(got["quotient_commitment"]), "quotient_evaluations": got["quotient_evaluations"],
          "opening": {"lr": [(V.aff(C, xy[0], li[0]), V.aff(C, xy[1], li[1])) for xy, li in got["opening"]["lr"]], "delta": V.aff(C, *got["opening"]["delta"]),
                      "z1": got["opening"]["z1"], "z2": got["opening"]["z2"], "sg": V.aff(C, *got["opening"]["sg"])}}
    for k in ("commitments", "quotient_commitment", "zeta_evaluations", "zeta_omega_evaluations"):
        assert pr[k] == want[k], k
    assert tuple(map(list, pr["quotient_evaluations"])) == tuple(map(list, want["quotient_evaluations"]))
    assert got["challenges"] == want["challenges"]
    assert pr["opening"] == want["opening"]
    assert O.verify(C, logn, osrs, cons, pr, P.StdRng(bytes([2] * 32)))
    bad = dict(pr, zeta_evaluations=[(pr["zeta_evaluations"][0] + 1) % F.p] + list(pr["zeta_evaluations"][1:]))
    assert not O.verify(C, logn, osrs, cons, bad, P.StdRng(bytes([2] * 32)))
    if which == "trace":                                          # an unsatisfied trace cannot be proved
        broken = dict(dinp, error=lim([1] + inp["error"][1:]))
        with pytest.raises(RuntimeError):
            o1vm.prove(cid, logn, srs, broken, cons, V.RefRng(P.StdRng(seed)))
