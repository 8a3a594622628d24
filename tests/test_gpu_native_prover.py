"""kh_prove -- ProverProof::create as one native call (csrc/prover.cpp, the host loop in C++ over the library's own C ABI) -- against the
oracle's CPU prover (oracle/prover.py, pinned on the reference's whole-proof vector) and against the Python device prover, on the same
circuit, witness and random stream: the same proof, byte for byte.  Covers generic circuits with copy constraints and public inputs, both
curves, an SRS longer than the domain, chunked proofs (domain larger than the SRS), the whole gate library with an optional gate, the
all-gates mode, a device-resident witness, the library's own randomness, and the error paths."""
import random

import numpy as np
import pytest

from oracle import circuit as CC
from oracle import kimchi as K
from oracle import pasta as P
from oracle import prover as OPR
from oracle import views as V

from test_gpu_prover_parity import _limbs, compare, device_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def generic_circuit(F, logn, log_srs, rnd, npub=3):
    p = F.p
    n = 1 << logn
    nch = 1 << max(0, logn - log_srs)
    zk = (16 * nch + 5) // 7
    rows = n - zk - 5
    gates, wit = [], [[0] * rows for _ in range(15)]
    for r in range(rows):
        a, b = int(rnd.integers(1, 1 << 62)), int(rnd.integers(1, 1 << 62))
        if r < npub:
            gates.append(CC.generic_gadget(p, r, CC.generic_spec(p, "Pub")))
            wit[0][r] = a
        else:
            gates.append(CC.generic_gadget(p, r, CC.generic_spec(p, "Add"), CC.generic_spec(p, "Mul")))
            wit[0][r], wit[1][r], wit[2][r] = a, b, (a + b) % p
            wit[3][r], wit[4][r], wit[5][r] = b, a, a * b % p
    for r in range(npub, rows - 1, 2):
        CC.connect_cell_pair(gates, (r, 0), (r, 4))
    cs = CC.build(F, gates, public=npub, max_poly_size=1 << log_srs)
    assert cs["log2_n"] == logn and cs["zk_rows"] == zk
    return cs, wit


@pytest.mark.parametrize("cid,logn,log_srs", [(0, 7, 7), (1, 7, 7), (0, 8, 10), (0, 9, 7), (1, 8, 7)])
def test_native_proof_equals_the_oracle_provers_proof(khip, cid, logn, log_srs):
    from proof_systems_amd import prover
    C = P.CURVES[cid]; F = C.scalar
    cs, wit = generic_circuit(F, logn, log_srs, np.random.default_rng(300 * cid + logn))
    CC.verify_witness(cs, wit)
    seed = bytes([17 + logn, cid] + [9] * 30)
    osrs = OPR.Srs(C, 1 << log_srs)
    oix = OPR.Index(C, cs, osrs)
    oproof = OPR.create_proof(oix, wit, P.StdRng(seed))
    srs = khip.Srs.create(cid, 1 << log_srs)
    ix = device_index(khip, cs, cid, srs)
    w = np.stack([_limbs(F, col) for col in wit])
    nproof = prover.create_proof_native(ix, w, V.RefRng(P.StdRng(seed)))
    c, vix, pr = V.device_views(ix, nproof)
    assert compare(C, oproof, pr) is None, compare(C, oproof, pr)
    assert OPR.serialize_proof(C, pr) == OPR.serialize_proof(C, oproof)
    # ... and the Python device prover's, challenge for challenge
    dproof = prover.create_proof(ix, w, V.RefRng(P.StdRng(seed)))
    assert dproof["challenges"] == nproof["challenges"]
    assert V.device_views(ix, dproof)[2] == pr
    ix.free()


def library_circuit(F, rnd, optional=("ForeignFieldAdd",)):
    """generic rows + one instance of every library gate + optional gates that need no lookup table"""
    from test_gates import gate_rows, tables
    wrows, crows, types = [], [], []
    for r in range(6):
        wrows.append([5] + [0] * 14); crows.append([1, 0, 0, 0, F.p - 5] + [0] * 10); types.append("Generic")
    for name in ("Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar") + tuple(optional):
        w, co, ngate = tables(name, rnd)
        live = set(gate_rows(name, ngate))
        for r, (wr, cr) in enumerate(zip(w, co)):
            wrows.append(list(wr)); crows.append([c % F.p for c in cr]); types.append(name if r in live else "Zero")
    return wrows, crows, types


def test_native_proof_over_the_gate_library(khip):
    """every always-present gate type and an optional one live in one circuit: native = Python device prover (same challenges, same proof),
    accepted by the oracle verifier; all-gates mode gives the same proof; a broken row fails the zero-remainder check with the reference's message"""
    from proof_systems_amd import prover
    from test_gpu_prover import _verify
    rnd = random.Random(77)
    F = prover.Fld(khip.FP)
    wrows, crows, types = library_circuit(P.Fp, rnd)
    rows = len(wrows)
    logn = 7
    assert rows + 3 <= 1 << logn
    ix = prover.ProverIndex(khip.VESTA, logn, np.stack([F.limbs_many(r) for r in crows]), gate_types=types)
    assert ix.optional == ["ForeignFieldAdd"]
    wit = np.stack([F.limbs_many([wrows[r][c] for r in range(rows)]) for c in range(15)])
    dproof = prover.create_proof(ix, wit, np.random.default_rng(12))
    nproof = prover.create_proof_native(ix, wit, np.random.default_rng(12))
    assert nproof["challenges"] == dproof["challenges"]
    assert V.device_views(ix, nproof)[2] == V.device_views(ix, dproof)[2]
    assert _verify(khip, ix, nproof)[0]
    for key in ("poseidon_selector", "complete_add_selector", "mul_selector", "emul_selector", "endomul_scalar_selector"):
        assert nproof["evals"][key][0][0] != 0
    assert nproof["evals"]["optional_gate_selectors"][2] is not None and nproof["evals"]["optional_gate_selectors"][0] is None
    aproof = prover.create_proof_native(ix, wit, np.random.default_rng(12), all_gates=True)
    assert V.device_views(ix, aproof)[2] == V.device_views(ix, nproof)[2]
    tm = {}
    prover.create_proof_native(ix, wit, np.random.default_rng(13), timings=tm, check=False)
    assert set(tm) == set(khip.PROOF_PHASES) | {"total"} and all(v > 0 for v in tm.values())
    r0 = types.index("Poseidon") + 3
    wrows[r0][7] = (wrows[r0][7] + 1) % F.p
    bad = np.stack([F.limbs_many([wrows[r][c] for r in range(rows)]) for c in range(15)])
    with pytest.raises(khip.KhError, match="vanishing polynomial"):
        prover.create_proof_native(ix, bad, np.random.default_rng(12))
    # KH_PROVE_EAGER_CHECK: the same error, raised at the quotient phase (prover.rs:913-917) instead of after the opening; a good witness is unaffected
    import time
    t0 = time.perf_counter()
    with pytest.raises(khip.KhError, match="vanishing polynomial"):
        prover.create_proof_native(ix, bad, np.random.default_rng(12), eager_check=True)
    t_eager = time.perf_counter() - t0
    t0 = time.perf_counter()
    with pytest.raises(khip.KhError, match="vanishing polynomial"):
        prover.create_proof_native(ix, bad, np.random.default_rng(12))
    t_deferred = time.perf_counter() - t0
    assert t_eager < t_deferred, (t_eager, t_deferred)             # no opening behind the failed check
    eproof = prover.create_proof_native(ix, wit, np.random.default_rng(12), eager_check=True)
    assert V.device_views(ix, eproof)[2] == V.device_views(ix, nproof)[2]
    ix.free()


def test_native_prover_draws_its_own_randomness_and_takes_a_resident_witness(khip):
    from proof_systems_amd import prover
    from test_gpu_prover import _verify
    ix = prover.bench_circuit_index(khip.VESTA, 10)
    F = ix.F
    n, zk = ix.n, ix.zk_rows
    wit = np.zeros((15, n - 10, 4), dtype=np.uint64); wit[0, :, :] = F.limbs(1)
    p1 = prover.create_proof_native(ix, wit, None)                       # getrandom
    p2 = prover.create_proof_native(ix, wit, None)
    assert _verify(khip, ix, p1)[0] and _verify(khip, ix, p2)[0]
    assert p1["challenges"]["zeta"] != p2["challenges"]["zeta"]          # blinded: two proofs of the same statement differ
    # the padded columns already on the device (the caller has randomised the zero-knowledge rows)
    rng = np.random.default_rng(5)
    full = np.zeros((15, n, 4), dtype=np.uint64); full[:, :n - 10, :] = wit
    full[:, n - zk:, :] = F.limbs_many(F.rand_many(rng, 15 * zk)).reshape(15, zk, 4)
    dev = khip.DevBuf(15 * n * 32).upload(full)
    a = prover.create_proof_native(ix, None, np.random.default_rng(6), witness_on_device=dev)
    b = prover.create_proof(ix, None, np.random.default_rng(6), witness_on_device=dev)
    assert V.device_views(ix, a)[2] == V.device_views(ix, b)[2] and _verify(khip, ix, a)[0]
    nx = prover.native_index(ix)
    with pytest.raises(khip.KhError, match="random elements"):
        nx.prove(witness=wit, randomness=np.zeros((3, 4), np.uint64))
    with pytest.raises(khip.KhError, match="NoRoomForZkInWitness"):
        nx.prove(witness=np.zeros((15, n, 4), np.uint64))
    dev.free(); ix.free()


def test_native_provers_on_several_threads(khip):
    """kh_prove from four host threads at once -- two on ONE index / SRS handle (openings on a handle queue), two on their own --, randomness from
    the library: every proof is accepted, and a seeded proof made while the others run equals the one made alone."""
    import threading
    from proof_systems_amd import prover
    from test_gpu_prover import _verify
    ixs = [prover.bench_circuit_index(khip.VESTA, 10) for _ in range(3)]
    F = ixs[0].F
    n = ixs[0].n
    wit = np.zeros((15, n - 10, 4), dtype=np.uint64); wit[0, :, :] = F.limbs(1)
    alone = prover.create_proof_native(ixs[0], wit, np.random.default_rng(77))
    out, errs = {}, []

    def work(t, ix, seeded):
        try:
            for r in range(4):
                out[(t, r)] = prover.create_proof_native(ix, wit, np.random.default_rng(77) if seeded else None, check=(r == 0))
        except BaseException as e:                                  # noqa: BLE001 -- reported by the main thread
            errs.append(e)
    th = [threading.Thread(target=work, args=(0, ixs[0], True)), threading.Thread(target=work, args=(1, ixs[0], False)),
          threading.Thread(target=work, args=(2, ixs[1], False)), threading.Thread(target=work, args=(3, ixs[2], False))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    assert len(out) == 16
    for r in range(4):
        assert V.device_views(ixs[0], out[(0, r)])[2] == V.device_views(ixs[0], alone)[2]
    for t, ix in ((1, ixs[0]), (2, ixs[1]), (3, ixs[2])):
        assert _verify(khip, ix, out[(t, 3)])[0]
    for ix in ixs:
        ix.free()


def test_every_seeded_proof_under_four_saturating_provers_is_the_one_made_alone(khip):
    """Four threads, each on its own index and its own private context (kh_prove's default), 2^14 gates so that the GPU is saturated and
    kernels of different provers queue behind each other: EVERY proof is compared with the seeded proof made alone (a stale or early read
    anywhere in the pipeline -- the halves of sg against the last fold was one -- shows up as a differing component).  Then the same with
    KH_PROVE_SHARED_CONTEXT (all four on the library's one context)."""
    import threading
    from proof_systems_amd import prover
    T, reps, logn = 4, 60, 14
    ixs = [prover.bench_circuit_index(khip.VESTA, logn) for _ in range(T)]
    F = ixs[0].F
    wit = np.tile(F.limbs(1), (15, (1 << logn) - 10, 1))
    ref = [V.device_views(ixs[t], prover.create_proof_native(ixs[t], wit, np.random.default_rng(100 + t)))[2] for t in range(T)]
    for shared in (False, True):
        bad, errs = [], []

        def work(t):
            try:
                for i in range(reps if not shared else reps // 3):
                    pr = V.device_views(ixs[t], prover.create_proof_native(ixs[t], wit, np.random.default_rng(100 + t), check=False, shared_context=shared))[2]
                    if pr != ref[t]:
                        bad.append((t, i, [k for k in pr if pr[k] != ref[t][k]]))
            except BaseException as e:                              # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join(timeout=600)
        assert not errs, errs
        assert not bad, (shared, bad[:5])
    for ix in ixs:
        ix.free()


def test_a_c_program_proves_and_the_oracle_verifier_accepts(khip, tmp_path):
    """tests/cpp/test_prove.cpp: index columns, proof and randomness through the C ABI alone (no Python in the process); the proof it writes
    verifies against the verifier index of the same circuit built here -- the recipe a Rust shim follows (rust/kimchi-hip/src/prover.rs)."""
    import os
    import subprocess
    from proof_systems_amd import prover
    from test_gpu_prover import _verify
    logn = 10
    ix = prover.bench_circuit_index(khip.VESTA, logn)
    F = ix.F; nch = ix.num_chunks
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "proof.bin")
    np.concatenate([np.array([logn], dtype=np.uint64), F.limbs_many(ix.shifts).reshape(-1), np.asarray(ix.digest, dtype=np.uint64).reshape(-1)]).tofile(inp)
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_prove")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PROVE_OK" in r.stdout, r.stdout + r.stderr
    raw = np.fromfile(outp, dtype=np.uint8)
    pos, sec = 0, []
    for _ in range(12):
        cnt, pts = (int(x) for x in raw[pos:pos + 16].view(np.uint64)); pos += 16
        w = 8 if pts else 4
        limbs = raw[pos:pos + 8 * w * cnt].view(np.uint64).reshape(cnt, w).copy(); pos += 8 * w * cnt
        flags = raw[pos:pos + cnt].copy() if pts else None
        pos += cnt if pts else 0
        sec.append((limbs, flags))
    assert pos == raw.size
    comms = lambda s, k: [(sec[s][0][i * nch:(i + 1) * nch], sec[s][1][i * nch:(i + 1) * nch]) for i in range(k)]
    ev = F.values(sec[4][0])
    E = [(ev[(2 * j) * nch:(2 * j + 1) * nch], ev[(2 * j + 1) * nch:(2 * j + 2) * nch]) for j in range(len(ev) // (2 * nch))]
    pe = F.values(sec[5][0]); z12 = F.values(sec[9][0])
    proof = {"w_comm": comms(0, 15), "z_comm": comms(1, 1)[0], "t_comm": sec[2], "public_comm": comms(3, 1)[0], "ft_eval1": F.values(sec[6][0])[0], "prev_challenges": [],
             "evals": {"public": (pe[:nch], pe[nch:]), "z": E[0], "generic_selector": E[1], "poseidon_selector": E[2], "complete_add_selector": E[3], "mul_selector": E[4],
                       "emul_selector": E[5], "endomul_scalar_selector": E[6], "w": E[7:22], "coefficients": E[22:37], "s": E[37:43], "optional_gate_selectors": [None] * 6},
             "opening": {"lr": [(sec[7][0][2 * r:2 * r + 2], sec[7][1][2 * r:2 * r + 2]) for r in range(sec[7][0].shape[0] // 2)], "delta": (sec[8][0][0], bool(sec[8][1][0])),
                         "z1": z12[0], "z2": z12[1], "sg": (sec[10][0][0], bool(sec[10][1][0]))}}
    assert len(E) == 43
    assert _verify(khip, ix, proof)[0]
    ix.free()


@pytest.mark.parametrize("seed", range(6))
def test_randomised_configurations_native_equals_python_and_verifies(khip, seed):
    """Differential fuzz over what the parity tests fix by hand: a random subset of the library / optional gates (in random order, between random
    numbers of generic rows), a random number of public inputs, the domain 1x / 2x / 4x the SRS or half of it -- kh_prove and the Python loop give the
    same proof from the same random stream, and the oracle verifier accepts it."""
    from proof_systems_amd import prover
    from test_gates import gate_rows, tables
    from test_gpu_prover import _verify
    rnd = random.Random(9000 + seed)
    F = prover.Fld(khip.FP)
    names = ["Poseidon", "CompleteAdd", "VarBaseMul", "EndoMul", "EndoMulScalar", "ForeignFieldAdd", "Rot64"]
    rnd.shuffle(names)
    names = names[:rnd.randrange(0, len(names) + 1)]
    npub = rnd.randrange(0, 4)
    wrows, crows, types = [], [], []

    def generic_rows(k, public=False):
        for _ in range(k):
            v = rnd.randrange(1, F.p)
            wrows.append([v] + [0] * 14)
            crows.append([1, 0, 0, 0, 0 if public else (F.p - v)] + [0] * 10)           # public rows: w0 - p_i = 0 through the public polynomial
            types.append("Generic")
    generic_rows(npub, public=True)
    generic_rows(rnd.randrange(1, 6))
    for name in names:
        w, co, ngate = tables(name, rnd)
        live = set(gate_rows(name, ngate))
        for r, (wr, cr) in enumerate(zip(w, co)):
            wrows.append(list(wr)); crows.append([c % F.p for c in cr]); types.append(name if r in live else "Zero")
        generic_rows(rnd.randrange(0, 3))
    rows = len(wrows)
    logn = 7 if rows + 9 <= 128 else 8
    log_srs = logn + rnd.choice([-2, -1, 0, 0, 1])
    nch = 1 << max(0, logn - log_srs)
    assert rows + (16 * nch + 5) // 7 <= 1 << logn
    srs = khip.Srs.create(khip.VESTA, 1 << log_srs)
    ix = prover.ProverIndex(khip.VESTA, logn, np.stack([F.limbs_many(r) for r in crows]), srs=srs, gate_types=types, public=npub)
    assert ix.num_chunks == nch
    wit = np.stack([F.limbs_many([wrows[r][c] for r in range(rows)]) for c in range(15)])
    a = prover.create_proof(ix, wit, np.random.default_rng(seed))
    b = prover.create_proof_native(ix, wit, np.random.default_rng(seed))
    assert a["challenges"] == b["challenges"], (names, npub, logn, log_srs)
    assert V.device_views(ix, a)[2] == V.device_views(ix, b)[2]
    assert _verify(khip, ix, b)[0], (names, npub, logn, log_srs)
    c = prover.create_proof_native(ix, wit, np.random.default_rng(seed), all_gates=True)
    assert V.device_views(ix, c)[2] == V.device_views(ix, b)[2]
    ix.free()
