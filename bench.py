#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Kimchi hot path.

Workload at N=1 (BASELINE.json configs[1]): ONE 2^20-point Pippenger MSM over the Vesta SRS
(bases = SRS::<Vesta>::create(1<<20).g, generated on the device by kh_srs_create_device;
scalars = uniform 254-bit Fp Montgomery limbs from a fixed-seed PRNG), inputs resident in HBM
when the timed region starts.  A "step" is one such MSM through the C ABI (digits -> sort ->
bucket accumulation -> reduction -> host finish); the timed loop keeps two of them in flight (KH_BENCH_DEPTH; round 4: two beat three and four at 20 and at 60 steps)
(kh_msm_submit / kh_msm_wait); the warm-up steps are synchronous and give the latency, five more
synchronous steps after the timed loop give the per-phase HIP-event timings and the dominant
kernel's own duration at the clocks the loop ran at.
`value` = Mscalar/s (whole job).  For N>1 (one process per GPU, RCCL) rank r owns the bases
g[r*2^20 .. (r+1)*2^20) of a (N*2^20)-point MSM (point-range sharding): every step
each rank reduces its slice, the partial sums are all-gathered (N x 72 bytes) and folded on
every rank (N group additions on the host, kh_points_sum).  Weak scaling.

Extra objects on the JSON line:
  roofline      dominant kernel (k_accumulate): algorithmic bytes (96 B per point-scalar pair,
                SURVEY 8d) / its HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline  the oracle's C Pippenger (oracle/pasta_ref.c, "port") on this box's host cores,
                same bases and scalars, on rank 0 at N=1; its result also cross-checks the
                GPU result bit-for-bit at full size
  ntt_kernels   (N=1) the transform half of the metric: iNTT 2^16 x 19 and LDE 2^16 -> 2^19 x 16 (the shapes of one proof), device
                resident, HIP-event timed: ms, algorithmic GB/s (64 B per element in place, 288 n per LDE column),
                fraction of the HBM peak, and the VALU-issue fraction from the PMC profile of the same build
  prover        (N=1) BASELINE config 3: ProverProof::create at 2^16 gates on Vesta -- a complete, verified proof by the
                device-resident pipeline (proof_systems_amd/prover.py) -> constraints/s; `seconds_all_gates` with every
                always-present gate type evaluated over d8 as the reference does; `cpu_baseline` = config 3's operation list
                on the host cores with the C port; the reference's own call pattern against the library (`dropin`: 15 threads
                on host buffers, 15 interpolations, 16 extensions); `pair` = BASELINE config 5 (Pallas + Vesta, 2^16 each)
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
ALG_BYTES_PER_PAIR = 96          # 64 B affine point + 32 B scalar, each read once
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_FILE = "r06_msm20_pmc.json"                 # tools/profile_msm.py (rocprofv3 PMC passes), keyed by the hash of csrc/
MIX_FILE = "r06_k_acc_wide29_valu_mix.json"   # tools/valu_mix.py (static opcode histogram of the loop body)
RATES_FILE = "r04_valu_rates.json"              # per-opcode issue cycles measured by tools/microbench.hip
NTT_PMC_FILE = "r06_ntt_pmc.json"               # tools/profile_msm.py --workload ntt (tools/bench_ntt.py --bench-shapes)
NTT_MIX_FILE = "r06_k_ntt_pass_valu_mix.json"   # tools/valu_mix.py --kernel ntt
BUSY_FILE = "r06_proof_busy.json"               # tools/proof_busy.py (rocprofv3 kernel trace of one 2^16 proof): kernel time / wall


def rand_scalars(rng, n):
    s = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 62) - 1)          # < 2^254 < p: valid Montgomery limbs
    return s


def ntt_block(khip):
    """The transform shapes of one 2^16 proof, device resident (kh_ntt_dev / kh_lde_dev), timed with the library's own HIP events
    on its stream (khip.last_timings).  Algorithmic bytes: SURVEY 8d."""
    rng = np.random.default_rng(7)

    def timed(fn, reps=10):
        fn(); khip.sync()
        ts = []
        for _ in range(reps):
            fn(); khip.sync()
            ts.append(sum(ms for _, ms in khip.last_timings()))
        return float(np.median(ts))
    n = 1 << 16
    out = {}
    buf = khip.DevBuf(19 * n * 32).upload(rand_scalars(rng, 19 * n))
    ms = timed(lambda: khip.ntt_dev(khip.FP, buf, 16, True, 19))
    gb = 64.0 * n * 19 / (ms * 1e-3) / 1e9
    out["intt_2^16_x19"] = {"ms": ms, "algorithmic_GBps": gb, "hbm_frac": gb / HBM_PEAK_GBS, "algorithmic_bytes": 64 * n * 19}
    buf.free()
    src = khip.DevBuf(16 * n * 32).upload(rand_scalars(rng, 16 * n)); dst = khip.DevBuf(16 * 8 * n * 32)
    ms = timed(lambda: khip.lde_dev(khip.FP, src, 16, 3, dst, 16))
    gb = 288.0 * n * 16 / (ms * 1e-3) / 1e9
    out["lde_2^16_to_2^19_x16"] = {"ms": ms, "algorithmic_GBps": gb, "hbm_frac": gb / HBM_PEAK_GBS, "algorithmic_bytes": 288 * n * 16}
    src.free(); dst.free()
    out["bound"] = "integer ALU (VALU issue): 0.41 log2 N + 1 Montgomery products per element; DESIGN.md section 4"
    here = source_hash()
    pmc = load_profile(NTT_PMC_FILE); mix = load_profile(NTT_MIX_FILE); rates = load_profile(RATES_FILE)
    key = next((k for k in (pmc or {}).get("kernels", {}) if k.startswith("k_ntt_pass<FpParams")), None)
    if not pmc or pmc.get("source_sha256") != here or key is None:
        out["valu_issue_note"] = "profiles/%s was not collected on this build: counter-derived fields withheld" % NTT_PMC_FILE
        return out
    k = pmc["kernels"][key]
    per_instr = None
    if mix and rates and mix.get("source_sha256") == here:
        hist = mix["valu_histogram"]; tot = float(sum(hist.values()))
        per_instr = sum(c * rates["cycles"].get(o.replace("_e32", "").replace("_e64", ""), rates["default_cycles"]) for o, c in hist.items()) / tot
    if per_instr and k.get("sustained_clock_ghz"):
        issue_ms = k["SQ_INSTS_VALU"] * per_instr / 1024.0 / (k["sustained_clock_ghz"] * 1e9) * 1e3
        out["valu_issue"] = {"kernel": key, "instructions_per_launch": k["SQ_INSTS_VALU"], "issue_cycles_per_instruction": per_instr, "avg_launch_ms": k["avg_ns"] * 1e-6,
                             "frac": issue_ms * k["sustained_clock_ghz"] / 2.4 / (k["avg_ns"] * 1e-6),
                             "sustained_clock_ghz": k["sustained_clock_ghz"], "frac_at_sustained_clock": issue_ms / (k["avg_ns"] * 1e-6),
                             "mix": "static opcode histogram of the whole kernel (profiles/%s), per-opcode cycles from profiles/%s" % (NTT_MIX_FILE, RATES_FILE),
                             "traffic_bytes_per_launch": k.get("fetch_raw_bytes", 0.0) + k.get("write_bytes", 0.0), "source": "profiles/" + NTT_PMC_FILE,
                             "workload": pmc.get("command")}
    return out


def cpu_quota():
    """CPUs this container may actually use at once (cgroup v2 cpu.max / v1 cfs quota; None = no limit): os.cpu_count() reports the host's threads
    whatever the quota is, and a CPU baseline on `cores` threads under a smaller quota is throttled, not parallel."""
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p_)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p_
    except (OSError, ValueError):
        return None


def usable_threads():
    """threads for the CPU baselines: the host's hardware threads, the affinity mask and the cgroup quota, whichever is smallest (256 threads
    under a 16-CPU quota are throttled in bursts: slower than 16)"""
    n = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    q = cpu_quota()
    return max(1, min(n, int(q + 0.999))) if q else n


def prover_cpu_baseline(khip, ix, wit_padded, log_n=16):
    """BASELINE config 3's operation list (15 Lagrange-basis MSMs of the benchmark witness + 1 + 7 monomial MSMs + the 32
    round MSMs of the opening, 19 iNTT(2^16) + 16 LDE(-> 2^19) + iNTT(2^18) + iNTT(2^19)) on THIS box's host cores with the C port
    (oracle/pasta_ref.c: signed-window Pippenger with (window, point-slice) jobs from a shared queue, radix-2 NTT parallel over
    columns or, for fewer columns than threads, over the butterflies of a stage) -- `kind: "port"`: the Rust reference cannot be
    built here.  It is the data-parallel part of ProverProof::create only (no gate evaluation, no transcript), i.e. a LOWER bound on
    what a CPU prover of this family needs on this host.  Never the target; checker-side code, outside every timed region."""
    from oracle import cref
    n = 1 << log_n
    cores = os.cpu_count() or 1
    thr = usable_threads()                                                     # every CPU this container may use (oracle/pasta_ref.c schedules its jobs over them: pick_schedule, ntt_flat_worker)
    rng = np.random.default_rng(5)
    g = ix.srs.get_g()
    lag, linf = ix.srs.get_lagrange(log_n)
    uni = rand_scalars(rng, n)
    t0 = time.perf_counter()
    for i in range(15):                                                       # witness commitments (mostly ones: the bench circuit)
        cref.msm(0, lag, wit_padded[i], inf=linf, threads=thr)
    for _ in range(8):                                                        # z + the seven chunks of t (uniform coefficients)
        cref.msm(0, g, uni, threads=thr)
    m = n // 2
    while m >= 1:                                                             # the opening: L and R over half of the current basis
        for _ in range(2):
            cref.msm(0, g[:m], uni[:m], threads=thr)
        m //= 2
    t_msm = time.perf_counter() - t0
    cols = rand_scalars(rng, 19 * n).reshape(19, n, 4)
    c19, c16 = cols.copy(), cols[:16].copy()                                  # inputs made before the clock starts
    q4, q8 = rand_scalars(rng, 4 * n).reshape(1, 4 * n, 4), rand_scalars(rng, 8 * n).reshape(1, 8 * n, 4)
    t0 = time.perf_counter()
    cref.ntt(0, c19, log_n, True, threads=thr)
    cref.lde(0, c16, log_n, 3, threads=thr)
    cref.ntt(0, q4, log_n + 2, True, threads=thr)
    cref.ntt(0, q8, log_n + 3, True, threads=thr)
    t_ntt = time.perf_counter() - t0
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"seconds": t_msm + t_ntt, "msm_seconds": t_msm, "ntt_seconds": t_ntt, "constraints_per_s": n / (t_msm + t_ntt), "cores": thr, "host_cores": cores, "cgroup_cpu_quota": cpu_quota(), "usable_cpus": len(os.sched_getaffinity(0)),
            "cpu_model": model, "kind": "port",
            "sample": "the whole operation list of one 2^16 proof once: 23 commitment MSMs + 32 opening-round MSMs, 19 iNTT(2^16), 16 LDE(2^16 -> 2^19), iNTT(2^18), iNTT(2^19)",
            "note": "data-parallel part only (no constraint evaluation, no sponge): a lower bound for a CPU prover of this algorithmic family on this host; not the reference binary"}


def prover_block(khip, srs20, check_with_oracle=True, log_n=16, reps=5):
    """BASELINE config 3: ProverProof::create for the benchmark circuit of kimchi/src/bench.rs at 2^16 gates on Vesta -- a
    COMPLETE proof (real data flow, real Fiat-Shamir challenges from the native sponges, zero remainder asserted), timed from a host
    witness through kh_prove, the library's own host loop in C++ (`seconds`, `phases_s`, `seconds_all_gates`), and through the Python
    loop over the same C-ABI calls (`python_loop`: also with the witness already in HBM, `seconds_resident`).  The proof of the last repetition is handed to
    the oracle's restatement of the reference VERIFIER (checker leg, outside every timed region).  `dropin` times the
    reference's own call pattern against the library: 15 host threads calling commit_evaluations_non_hiding on host
    buffers at once (prover.rs:329-351), one interpolate per column on host buffers (prover.rs:370-381)."""
    import threading
    from proof_systems_amd import prover
    n = 1 << log_n
    khip.set_phase_timers(False)                                         # nothing in this block reads per-phase HIP events (the provers time themselves on the host clock)
    g16 = srs20.get_g(0, n)
    srs16 = khip.Srs(khip.VESTA, g16)
    t0 = time.perf_counter()
    ix = prover.bench_circuit_index(khip.VESTA, log_n, srs=srs16)       # Lagrange basis (device group-iNTT), column forms, index commitments
    t_index = time.perf_counter() - t0
    F = prover.Fld(khip.FP)
    wit = np.tile(F.limbs(1), (15, n - 10, 1))                           # kimchi/src/bench.rs:106
    rng = np.random.default_rng(2024)
    prover.create_proof(ix, wit, rng)                                    # warm-up (workspaces, hipGraph capture of the round MSM)
    best, proof = None, None
    for _ in range(reps):
        t = {}
        khip.sync()
        proof = prover.create_proof(ix, wit, rng, timings=t)
        if best is None or t["total"] < best["total"]:
            best = t
    padded = np.zeros((15, n, 4), dtype=np.uint64); padded[:, :n - 10] = wit
    padded[:, n - 3:] = F.limbs_many([F.rand(rng) for _ in range(45)]).reshape(15, 3, 4)
    d_w = khip.DevBuf(padded.nbytes).upload(padded)
    best_res = None
    for _ in range(reps):
        t = {}
        khip.sync()
        prover.create_proof(ix, None, rng, timings=t, witness_on_device=d_w)
        if best_res is None or t["total"] < best_res["total"]:
            best_res = t
    d_w.free()
    prover.create_proof(ix, wit, rng, all_gates=True)
    best_all = None
    for _ in range(3):
        t = {}
        khip.sync()
        prover.create_proof(ix, wit, rng, timings=t, all_gates=True)
        if best_all is None or t["total"] < best_all["total"]:
            best_all = t
    # the same proof through kh_prove: the host loop in C++ (csrc/prover.cpp) -- what a Rust / C caller of the library pays
    prover.create_proof_native(ix, wit, rng)
    best_nat, best_nat_all = None, None
    for _ in range(reps):
        t = {}
        khip.sync()
        nproof = prover.create_proof_native(ix, wit, rng, timings=t)
        if best_nat is None or t["total"] < best_nat["total"]:
            best_nat = t
    for _ in range(3):
        t = {}
        khip.sync()
        prover.create_proof_native(ix, wit, rng, timings=t, all_gates=True)
        if best_nat_all is None or t["total"] < best_nat_all["total"]:
            best_nat_all = t
    out = {"workload": "ProverProof::create, benchmark circuit (2^%d - 10 generic gates), Vesta, SRS 2^%d, one chunk" % (log_n, log_n),
           "entry": "kh_prove: the host loop in C++ over the C ABI (csrc/prover.cpp) -- what a Rust / C caller of the library pays; byte-identical to the oracle prover "
                    "(tests/test_gpu_native_prover.py, tests/test_gpu_prover_parity.py)",
           "seconds": best_nat["total"], "constraints_per_s": n / best_nat["total"], "phases_s": {k: v for k, v in best_nat.items() if k != "total"},
           "seconds_all_gates": best_nat_all["total"], "constraints_per_s_all_gates": n / best_nat_all["total"],
           "python_loop": {"entry": "proof_systems_amd.prover.create_proof: the same sequence of C-ABI calls driven from Python (all features incl. the A/B switches)",
                           "seconds": best["total"], "constraints_per_s": n / best["total"], "phases_s": {k: v for k, v in best.items() if k != "total"},
                           "seconds_resident": best_res["total"], "constraints_per_s_resident": n / best_res["total"],
                           "seconds_all_gates": best_all["total"], "constraints_per_s_all_gates": n / best_all["total"]},
           "all_gates_note": "Poseidon, CompleteAdd, VarBaseMul, EndoMul, EndoMulScalar constraints evaluated over d8 although their selectors are zero for this circuit, "
                             "and all 15 witness columns extended -- the work the reference does on every proof (prover.rs:824-868); same proof",
           "index_time_s": t_index,
           "reference_readme_seconds": 6.3, "reference_note": "o1-labs README figure for 2^16 gates, hardware unspecified; not measured here (no Rust toolchain)",
           "note": "complete proof: 15 + 1 + 7 commitments, 16 + 2 iNTT, 16 LDE, generic + permutation rows, division by Z_H (zero remainder asserted), 43 x 2 evaluations, ft, "
                   "16 opening rounds; sponges native on the host; constraints of the five gate types whose selectors are zero for this circuit are not evaluated (they add 0)"}
    busy = load_profile(BUSY_FILE)
    if busy and busy.get("source_sha256") == source_hash():
        out["gpu_busy_frac"] = busy["gpu_busy_frac"]
        out["gpu_busy"] = {k: busy[k] for k in ("wall_us", "busy_us", "pre_opening", "opening", "note") if k in busy}
        out["gpu_busy"]["source"] = "profiles/" + BUSY_FILE
    else:
        out["gpu_busy_frac"] = None
        out["gpu_busy_note"] = "profiles/%s was not collected on this build: withheld" % BUSY_FILE
    if check_with_oracle and log_n == 16:
        out.update(fixture_parity(khip, ix, wit))
    # ---- the reference's call pattern, unchanged: host buffers, 15 concurrent callers
    cols = [np.ascontiguousarray(padded[i]) for i in range(15)]
    res = [None] * 15

    bar = threading.Barrier(16)
    done = threading.Barrier(16)

    def one(i):                                                         # threads are started up front (a rayon pool exists before the proof does)
        for _ in range(3):
            bar.wait()
            res[i] = srs16.commit_evaluations_non_hiding(log_n, cols[i])
            done.wait()
    th = [threading.Thread(target=one, args=(i,)) for i in range(15)]
    for x in th:
        x.start()
    best_c = None
    for _ in range(3):
        bar.wait()
        t0 = time.perf_counter()
        done.wait()
        dt = time.perf_counter() - t0
        best_c = dt if best_c is None else min(best_c, dt)
    for x in th:
        x.join()
    def threaded(fs, reps=10):                                           # the callables at once, one (pre-started) thread each; best of the repeats after a warm-up
        bar2 = threading.Barrier(len(fs) + 1); done2 = threading.Barrier(len(fs) + 1)

        def w(f):
            for _ in range(reps):
                bar2.wait(); f(); done2.wait()
        th2 = [threading.Thread(target=w, args=(f,)) for f in fs]
        for x in th2:
            x.start()
        ts = []
        for _ in range(reps):
            bar2.wait(); t0 = time.perf_counter(); done2.wait(); ts.append(time.perf_counter() - t0)
        for x in th2:
            x.join()
        return min(ts[3:])                                              # (the first repeats create the threads' copy streams and pin the fresh host pages)
    # interpolate: Evaluations::interpolate transforms the caller's Vec in place (ifft_in_place); one column per call, from ONE thread and from 15 at once
    # (prover.rs:370-381 is a par_iter over the 15 columns)
    icol = [c.copy() for c in cols]
    best_n = None
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(15):
            khip.ntt(khip.FP, icol[i], log_n, inverse=True, in_place=True)
        dt = time.perf_counter() - t0
        best_n = dt if best_n is None else min(best_n, dt)
    best_n_thr = threaded([(lambda i=i: khip.ntt(khip.FP, icol[i], log_n, inverse=True, in_place=True)) for i in range(15)])
    coeffs16 = np.ascontiguousarray(padded[:, :, :].copy()); coeffs16 = np.concatenate([coeffs16, coeffs16[:1]])     # 16 columns of n coefficients (15 w + z)
    outs16 = [np.ones((1, 8 << log_n, 4), np.uint64) for _ in range(16)]                                            # the destination Vecs exist (no first-touch faults in the timing)
    # (the host-buffer path has a slow mode early in a process -- the same 16 extensions took 9.8-11.7 ms right after start-up and 5.5-6.2 ms from the third
    #  measurement on, whatever ran in between: two untimed batched calls first)
    warm = np.ones((16, 8 << log_n, 4), np.uint64)
    for _ in range(2):
        khip.lde(khip.FP, coeffs16, log_n, 3, out=warm)
    del warm
    best_l = None
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(16):
            khip.lde(khip.FP, coeffs16[i:i + 1], log_n, 3, out=outs16[i])    # evaluate_over_domain_by_ref(d8) of one column: n up, 8n down
        dt = time.perf_counter() - t0
        best_l = dt if best_l is None else min(best_l, dt)
    # (24 repetitions: single repetitions of this call run 2x slow on busy hosts -- tools/latency/numa_extend.py -- and two of round 5's collections had all of 7 in that mode)
    best_l_thr = threaded([(lambda i=i: khip.lde(khip.FP, coeffs16[i:i + 1], log_n, 3, out=outs16[i])) for i in range(16)], reps=24)       # constraints.rs:488-494: a par_iter over w and z
    t_batch = None
    for _ in range(3):                                                   # (the first call of this shape sizes the slot's scalar workspace)
        t0 = time.perf_counter()
        batch = srs16.msm_batch(padded, basis=log_n)
        dt = time.perf_counter() - t0
        t_batch = dt if t_batch is None else min(t_batch, dt)
    same = all(np.array_equal(res[i][0][0], batch[0][i]) for i in range(15))
    out["dropin"] = {"commit_15_threads_host_buffers_s": best_c, "commit_one_batched_call_host_buffers_s": t_batch, "threads_match_batch": bool(same),
                     "interpolate_15_columns_15_threads_host_buffers_s": best_n_thr, "interpolate_15_columns_one_thread_host_buffers_s": best_n,
                     "extend_16_columns_2^16_to_2^19_16_threads_host_buffers_s": best_l_thr, "extend_16_columns_2^16_to_2^19_one_thread_host_buffers_s": best_l,
                     "swap_crates_only_total_s": best_c + best_n_thr + best_l_thr, "swap_crates_only_total_one_thread_transforms_s": best_c + best_n + best_l,
                     "note": "what a Rust prover gets by only swapping in GpuSrs / the ark-poly patch (rust/ark-poly-patch: interpolate -> kh_ntt, evaluate_over_domain_by_ref -> kh_lde, "
                             "the zero padding never crosses PCIe); witness commitments + interpolations + 8x extensions of one proof, called as the reference calls them (15 / 15 / 16 "
                             "rayon workers at once; round 5: uploads and downloads run on the callers' own copy streams outside the library lock) and, for comparison, the transforms "
                             "from one thread; PCIe-bound (pageable host memory)"}
    if check_with_oracle:
        out["proof_accepted_by_oracle_verifier"] = bool(oracle_verifies(khip, ix, proof))
        out["python_loop"]["proof_accepted_by_oracle_verifier"] = out["proof_accepted_by_oracle_verifier"]
        out["proof_accepted_by_oracle_verifier"] = bool(oracle_verifies(khip, ix, nproof))
        out["cpu_baseline"] = prover_cpu_baseline(khip, ix, padded, log_n)
    # several provers in flight (one host thread and one SRS handle each): a single proof is mostly latency chains, independent proofs overlap
    T, per = 4, 10                                       # (5 per thread until round 5: 20 proofs in ~0.1 s moved 165-207 proofs/s from run to run)
    ixs = [ix] + [prover.bench_circuit_index(khip.VESTA, log_n) for _ in range(T - 1)]
    for j in ixs[1:]:
        prover.create_proof(j, wit, np.random.default_rng(2), check=False)
    nxs = [prover.native_index(j) for j in ixs]
    bar = threading.Barrier(T + 1)

    def run(t):
        nxs[t].prove(witness=wit, randomness=None, flags=0)      # warm this thread's own library context (kh_prove works on a private one)
        bar.wait()
        for _ in range(per):                             # kh_prove, randomness from the library: no Python between the steps of a proof
            nxs[t].prove(witness=wit, randomness=None, flags=0)
        bar.wait()
    th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for t_ in th:
        t_.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t_ in th:
        t_.join()
    out["concurrent"] = {"provers_in_flight": T, "proofs_per_s": T * per / dt, "constraints_per_s": T * per * (1 << log_n) / dt,
                         "note": "throughput of independent proofs on one GPU through kh_prove (host loop in C++, one thread per prover); `seconds` above is the latency of one"}
    for j in ixs:
        j.free()
    khip.set_phase_timers(True)
    return out


def pair_block(khip, log_n=16, per=4, check_with_oracle=True):
    """BASELINE config 5: the Pallas + Vesta recursion pair, 2^16 gates each, both SRS (with their window tables and Lagrange bases)
    and all four kernel instantiations (MSM over Fq / Fp coordinates, NTT over Fp / Fq) resident in one process.  One host thread
    per curve; the handles live on devices 0 and 1 when two GPUs are visible, else both on device 0.  No inter-GPU traffic (SURVEY 8e:
    replicas only).  Reports each curve's proof latency alone and the pair's throughput with both in flight."""
    from proof_systems_amd import prover
    ndev = max(1, khip.device_count())
    devs = [0, 1 % ndev]
    n = 1 << log_n
    state = [None, None]
    res = {}
    bar = threading.Barrier(3)

    def run(k):
        cid = (khip.VESTA, khip.PALLAS)[k]
        khip.set_device(devs[k])                                         # HIP's current device is per-thread state
        ix = prover.bench_circuit_index(cid, log_n)
        F = prover.Fld(ix.fid)
        wit = np.tile(F.limbs(1), (15, n - 10, 1))
        rng = np.random.default_rng(90 + k)
        proof = prover.create_proof_native(ix, wit, rng)                 # kh_prove: the host loop in C++, so the two curves' threads do not share a GIL
        state[k] = (ix, proof)
        bar.wait()                                                       # 1: both warm
        if k == 0:                                                       # latencies alone, one curve after the other
            pass
        for phase in (0, 1):
            bar.wait()
            if phase == k:
                best = None
                for _ in range(3):
                    t = {}
                    prover.create_proof_native(ix, wit, rng, timings=t, check=False)
                    best = t["total"] if best is None else min(best, t["total"])
                res["seconds_alone_" + ("vesta", "pallas")[k]] = best
            bar.wait()
        bar.wait()                                                       # both in flight
        nx = prover.native_index(ix)
        for _ in range(per):
            nx.prove(witness=wit, randomness=None, flags=0)
        bar.wait()
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t_ in th:
        t_.start()
    bar.wait()
    for _ in (0, 1):
        bar.wait(); bar.wait()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t_ in th:
        t_.join()
    res.update({"workload": "Pallas + Vesta pair, benchmark circuit 2^%d gates each, both curves resident in one process" % log_n, "devices": devs,
                "devices_visible": ndev, "pairs_per_s": per / dt, "constraints_per_s": 2 * per * n / dt, "seconds_per_pair_both_in_flight": dt / per})
    if check_with_oracle:
        res["pallas_proof_accepted_by_oracle_verifier"] = bool(oracle_verifies(khip, *state[1]))
    for ix, _ in state:
        ix.free()
    return res


def make_lib_comm(khip, dist, rank, world):
    """kh_comm_init on every rank over ONE id: rank 0 draws it (kh_comm_unique_id), torch.distributed only carries the 128 bytes."""
    uid = [khip.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    return khip.Comm(world, rank, uid[0])


def other_collective_check(khip, dist, sm, lib_comm, partial, result, rank, world, coll_dev, timeout_s=60.0):
    """Runs the combine of `partial` through the collective the timed loop did NOT use (in-library RCCL if torch carried the loop, torch's nccl
    all_gather if KH_BENCH_COMM=lib did) and compares with `result`.  Returns (report, hung)."""
    import threading
    from proof_systems_amd import sharded
    rep = {"backend": "nccl-torch" if lib_comm is not None else "rccl-lib"}

    import torch
    cuda_dev = torch.cuda.current_device() if coll_dev == "cuda" else None

    def work():
        try:
            if cuda_dev is not None:
                torch.cuda.set_device(cuda_dev)        # the current device is per thread: a fresh thread would use cuda:0 on every rank
            t0 = time.perf_counter()
            comm = None if lib_comm is not None else make_lib_comm(khip, dist, rank, world)
            rep["init_s"] = time.perf_counter() - t0
            other = sharded.RankShardedMsm.__new__(sharded.RankShardedMsm)
            other.curve, other.dist, other.coll_device, other.rank, other.world = sm.curve, dist, coll_dev, rank, world
            other.comm, other.always_collective, other.engine, other.collective_backend = comm, True, sm.engine, None
            o, i = other.combine(*partial)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter(); other.combine(*partial); ts.append(time.perf_counter() - t0)
            rep["combine_us_median"] = 1e6 * float(np.median(ts))
            rep["ran"] = other.collective_backend
            rep["same_point_as_timed_collective"] = bool(bool(i[0]) == result[1] and (result[1] or np.array_equal(o[0], result[0])))
            if comm is not None:
                comm.free()
        except Exception as e:                        # noqa: BLE001 -- reported in the line, the measurement stands
            rep["error"] = "%s: %s" % (type(e).__name__, e)
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        rep["error"] = "timed out after %.0f s inside the collective" % timeout_s
        return rep, True
    return rep, False


def fixture_parity(khip, ix, wit):
    """Checker leg (never timed): the SAME index and witness the timed proofs used, proved once more through kh_prove with the random stream of
    tests/golden/proof_fixtures/bench_vesta_2_16.json (StdRng::from_seed, the oracle's restatement), serialised as rmp-serde writes a ProverProof
    and compared with the committed bytes -- the proof the oracle's CPU prover (pinned on the reference's whole-proof vector) makes for BASELINE
    config 3 at its own size.  Imports oracle/ for the RNG restatement and the serialiser only."""
    import hashlib
    from oracle import pasta as P
    from oracle import prover as OPR
    from oracle import views as V
    from proof_systems_amd import prover
    d = os.path.join(ROOT, "tests", "golden", "proof_fixtures")
    try:
        rec = json.load(open(os.path.join(d, "bench_vesta_2_16.json")))
        want = open(os.path.join(d, "bench_vesta_2_16.proof.bin"), "rb").read()
    except OSError:
        return {"byte_identical_to_oracle": None, "byte_parity_note": "tests/golden/proof_fixtures/bench_vesta_2_16.* not found"}
    proof = prover.create_proof_native(ix, wit, V.RefRng(P.StdRng(bytes.fromhex(rec["seed_hex"]))))
    got = OPR.serialize_proof(P.VESTA, V.device_views(ix, proof)[2])
    return {"byte_identical_to_oracle": bool(got == want), "byte_parity_fixture": "tests/golden/proof_fixtures/bench_vesta_2_16.proof.bin",
            "byte_parity_sha256": hashlib.sha256(got).hexdigest(), "byte_parity_fixture_sha256": rec["proof_sha256"],
            "byte_parity_note": "kh_prove on this run's index and witness with the fixture's StdRng stream: the %d serialised bytes of the ProverProof against the oracle prover's" % len(want)}


def oracle_verifies(khip, ix, proof):
    """Checker leg (never timed): oracle/kimchi.py restates the reference verifier; the final MSM runs in the C oracle."""
    from oracle import kimchi as K
    from oracle import pasta as P
    from oracle import views as V
    c, vix, pr = V.device_views(ix, proof)
    return K.verify(c, vix, pr, None, vix["h"], P.StdRng(bytes([9] * 32)), final_msm=V.final_msm_c(c, ix.srs.get_g(), ix.size))


def source_hash():
    """sha256 over csrc/ -- the PMC / instruction-mix files under profiles/ carry the hash of the sources they were
    collected on; numbers from a different build are not quoted."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "proof_systems_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cuh", ".hpp", ".inc", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def load_profile(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(p))
    except (OSError, ValueError):
        return None


def roofline_block(kname, acc_ms, n, log_n):
    """roofline of the dominant kernel from THIS run's kernel duration; counter-derived fields only from profiles/ files
    whose source hash equals the hash of the sources that are running."""
    alg_bytes = ALG_BYTES_PER_PAIR * n
    block = {"bound": "hbm", "kernel": kname, "achieved": (alg_bytes / (acc_ms * 1e-3) / 1e9) if acc_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": (alg_bytes / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if acc_ms else None, "traffic": None, "kernel_ms": acc_ms, "algorithmic_bytes": alg_bytes,
             "note": "integer-ALU bound (about 10 Montgomery products of 131 v_mad_u64_u32 each per point addition), see DESIGN.md section 3"}
    here = source_hash()
    pmc = load_profile(PMC_FILE)
    mix = load_profile(MIX_FILE)
    rates = load_profile(RATES_FILE)
    if log_n != LOG_N or not acc_ms:
        return block
    key = kname + "<FqParams>"
    if not pmc or pmc.get("source_sha256") != here or key not in pmc.get("kernels", {}):
        block["traffic_note"] = "profiles/%s was not collected on this build (or lacks %s): counter-derived fields withheld" % (PMC_FILE, key)
        return block
    k = pmc["kernels"][key]
    # k_accumulate's reads are 64-byte gathers, not a wide coalesced stream: the raw FETCH_SIZE already exceeds the known
    # gather + entry bytes, so the guide's 2x under-count correction is not applied to it
    block["traffic"] = k.get("fetch_raw_bytes", 0.0) + k.get("write_bytes", 0.0)
    block["traffic_source"] = "profiles/" + PMC_FILE
    if mix and rates and mix.get("source_sha256") == here and "SQ_INSTS_VALU" in k:
        hist = mix["valu_histogram"]; tot = float(sum(hist.values()))
        cyc = {o: rates["cycles"].get(o.replace("_e32", "").replace("_e64", ""), rates["default_cycles"]) for o in hist}
        per_instr = sum(hist[o] * cyc[o] for o in hist) / tot                  # issue cycles per executed VALU wave-instruction
        issue_cycles = k["SQ_INSTS_VALU"] * per_instr / 1024.0                  # per SIMD
        peak_ms = issue_cycles / 2.4e9 * 1e3
        # The clock: `frac` prices the issue cycles at the nominal 2.4 GHz against the kernel time measured LIVE in this run.  The profile's clock estimate
        # (GRBM_GUI_ACTIVE / kernel time of the counter collection, another run on another lease) is only consistent with THAT run's kernel time, so the
        # sustained-clock fraction is computed inside the profile (round 6: mixing it with the live time gave 1.07); what the live run itself says about its
        # clock is the floor below which the measured time would beat the issue bound.
        clk = k.get("sustained_clock_ghz"); prof_ms = k.get("avg_ns", 0.0) * 1e-6
        block["valu_issue"] = {"instructions_per_launch": k["SQ_INSTS_VALU"], "issue_cycles_per_instruction": per_instr,
                               "issue_bound_ms_at_2.4GHz": peak_ms, "frac": peak_ms / acc_ms,
                               "live_clock_floor_ghz": issue_cycles / (acc_ms * 1e-3) / 1e9,
                               "sustained_clock_ghz": clk, "profile_kernel_ms": prof_ms or None,
                               "frac_at_sustained_clock": (peak_ms * 2.4 / clk / prof_ms) if (clk and prof_ms) else None,
                               "unit": "fraction of the issue-bound time for this instruction mix (per-opcode cycles from tools/microbench.hip); frac = live kernel time at "
                                       "the nominal clock, frac_at_sustained_clock = the counter collection's own kernel time at the clock its counters give"}
    return block


class _DryShard:
    """--dry-run stand-in for a device shard: every MSM 'returns' the blinding base h (kh_srs_h: host code, no GPU)"""

    def __init__(self, khip, curve):
        self.h = khip.srs_h(curve)

    def msm_submit(self, ptr, n, k):
        return ("dry", k)

    def msm_wait(self, ticket):
        return np.tile(self.h, (ticket[1], 1)), np.zeros(ticket[1], np.uint8)

    def msm_batch_dev(self, ptr, n, k, mont=True):
        return np.tile(self.h, (k, 1)), np.zeros(k, np.uint8)

    def msm(self, scalars, mont=True):
        return self.h.copy(), False

    def close(self):
        pass


class _DryEngine:
    def __init__(self, khip):
        self.khip = khip

    def make_shard(self, curve, start, count):
        return _DryShard(self.khip, curve)

    def msm(self, shard, scalars, mont=True):
        return shard.h.copy(), False

    def points_sum(self, curve, xy, inf):
        out, oinf = self.khip.points_sum(curve, xy, inf)      # the product's fold (host code)
        return out, bool(oinf)

    def free_shard(self, shard):
        pass


def host_scalars_block(khip, srs, sc, n, steps, depth, want):
    """The same MSM from HOST scalars (pageable memory: what SRS::commit_non_hiding(&DensePolynomial) hands over, poly-commitment/src/ipa.rs:638-683) -- never
    `value`: pipelined through kh_msm_submit_host / kh_msm_wait (`depth` in flight, the upload of MSM i + 1 under the accumulation of MSM i; two distinct host
    buffers, as a caller's polynomials are), and one at a time through kh_msm (two half-range jobs with chunked uploads).  Median of three regions."""
    bufs = [sc, sc.copy()]
    out, inf = srs.msm(bufs[1])                            # warm: sizes the slots' scalar workspaces, pins nothing
    ok = (bool(inf) == bool(want[1])) and (bool(inf) or bool(np.array_equal(out, want[0])))
    vals = []
    for _ in range(3):
        q = []
        t0 = time.perf_counter()
        for i in range(steps):
            if len(q) >= depth:
                o, f = srs.msm_wait(q.pop(0))
                ok = ok and (bool(f[0]) == bool(want[1])) and (bool(f[0]) or bool(np.array_equal(o[0], want[0])))
            q.append(srs.msm_submit_host(bufs[i & 1]))
        while q:
            o, f = srs.msm_wait(q.pop(0))
            ok = ok and (bool(f[0]) == bool(want[1])) and (bool(f[0]) or bool(np.array_equal(o[0], want[0])))
        vals.append(n * steps / (time.perf_counter() - t0) / 1e6)
    lat = []
    for i in range(7):
        t0 = time.perf_counter(); srs.msm(bufs[i & 1]); lat.append(1e3 * (time.perf_counter() - t0))
    return {"value_host_scalars": sorted(vals)[1], "value_host_scalars_runs": vals, "ms_per_msm_host_scalars_synchronous": float(np.median(lat)),
            "host_scalars_results_match": bool(ok),
            "host_scalars_note": "PCIe-inclusive, pageable host memory, never `value`: %d MSMs per region through kh_msm_submit_host with %d in flight; synchronous = kh_msm" % (steps, depth)}


def gpu_locality(pci_bus_id):
    """(numa_node, cpus) of the PCIe device `pci_bus_id` ("0000:c1:00.0") from sysfs, or (None, None) where the kernel does not say."""
    base = "/sys/bus/pci/devices/%s/" % pci_bus_id.lower()
    try:
        node = int(open(base + "numa_node").read())
    except (OSError, ValueError):
        node = None
    cpus = None
    try:
        cpus = set()
        for part in open(base + "local_cpulist").read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
    except (OSError, ValueError):
        cpus = None
    return (node if node is not None and node >= 0 else None), (cpus or None)


def bind_rank_to_gpu(torch, device_index, world):
    """One process per GPU: keep this rank's host threads (submit loop, combiner, the library's helper thread, the runtime's staging copies) on the CPUs of
    the NUMA node its GPU hangs off -- a rank on the far socket pays the cross-socket hop on every doorbell, completion word and 72-byte partial.  Only at
    N > 1 (a lone rank keeps the scheduler's choice: that is how the N = 1 line has always been measured), only when sysfs names the locality and it
    intersects the CPUs the container may use; KH_BENCH_NO_AFFINITY=1 switches it off.  Returns what the rank banner prints."""
    info = {"pci": None, "numa_node": None, "cpus_bound": None}
    try:
        pr = torch.cuda.get_device_properties(device_index)
        info["pci"] = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:                                      # noqa: BLE001 -- diagnostics only
        return info
    node, cpus = gpu_locality(info["pci"])
    info["numa_node"] = node
    if world <= 1 or cpus is None or os.environ.get("KH_BENCH_NO_AFFINITY", "0") not in ("", "0"):
        return info
    mine = cpus & os.sched_getaffinity(0)
    if mine:
        try:
            os.sched_setaffinity(0, mine)
            info["cpus_bound"] = len(mine)
        except OSError:
            pass
    return info


class Watchdog:
    """A rank that hangs (a collective whose peer died, a kernel that never ends) must end the RUN, with a line that says where: every phase of main() arms a
    deadline; when one passes, this rank prints a JSON line with an `error` field (stdout, like the result line: the driver reads it instead of a time-out) and
    leaves with os._exit -- torch.distributed.run then takes the other ranks down.  KH_BENCH_WATCHDOG_S scales the deadlines (0 = off)."""

    def __init__(self, rank, world, base):
        self.rank, self.world, self.base = rank, world, base
        self.scale = float(os.environ.get("KH_BENCH_WATCHDOG_S", "1") or 0)
        self.deadline, self.name = None, None
        self.lock = threading.Lock()
        if self.scale > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def phase(self, name, seconds):
        with self.lock:
            self.name, self.deadline = name, (time.monotonic() + seconds * self.scale) if seconds else None

    def _run(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                name = self.name
            if late:
                emit_error(self.base, self.rank, self.world, "watchdog: phase '%s' exceeded its deadline on rank %d" % (name, self.rank))
                os._exit(4)


def emit_error(base, rank, world, msg):
    """ONE JSON line with an `error` field instead of a result (any rank may be the one that knows)"""
    line = dict(base)
    line.update({"value": None, "n_gpus": world, "error": msg, "error_rank": rank})
    sys.stdout.write(json.dumps(line) + "\n"); sys.stdout.flush()
    print("[bench rank %d/%d] ERROR: %s" % (rank, world, msg), file=sys.stderr, flush=True)


class Combiner:
    """The cross-rank combine of finished MSMs (all-gather of the 72-byte partial sums + local fold) on its OWN thread, in submission order: the thread that
    submits MSM i + 1 no longer waits inside the collective of MSM i.  Every rank pushes the same sequence, so the collectives match
    up rank to rank.  At N = 1 (no collective) the combine stays inline."""

    def __init__(self, sm, torch_mod, cuda_dev):
        import queue
        self.sm, self.q, self.last, self.err = sm, queue.Queue(), None, None
        self.torch, self.cuda_dev = torch_mod, cuda_dev
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        if self.cuda_dev is not None:
            self.torch.cuda.set_device(self.cuda_dev)      # the current device is per thread
        while True:
            item = self.q.get()
            try:
                if item is not None and self.err is None:
                    o, i = self.sm.combine(item[0], item[1])
                    self.last = (o[-1], bool(i[-1]))
            except Exception as e:                        # noqa: BLE001 -- re-raised on the submitting thread by drain()
                self.err = e
            finally:
                self.q.task_done()
            if item is None:
                return

    def push(self, xys, infs):
        self.q.put((xys, infs))

    def drain(self):
        self.q.join()
        if self.err is not None:
            raise self.err
        return self.last

    def close(self):
        self.q.put(None)
        self.th.join(5)


def main():
    base = {"metric": "MSM Mscalar/s at 2^%d (Vesta)" % LOG_N, "unit": "Mscalar/s", "higher_is_better": True}
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    try:
        _main(base)
    except SystemExit:
        raise
    except BaseException as e:                             # noqa: BLE001 -- a failed rank ends the run with a line that says why
        import traceback
        traceback.print_exc()
        emit_error(base, rank, world, "%s: %s" % (type(e).__name__, e))
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)


def _main(base):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="weak scaling: points PER GPU; with --strong: points in total")
    ap.add_argument("--strong", action="store_true", help="BASELINE config 4: ONE MSM of 2^log-n points (default 2^22) sharded over the ranks")
    ap.add_argument("--curve", choices=["vesta", "pallas"], default="vesta")
    ap.add_argument("--no-pipeline", action="store_true", help="time synchronous MSMs (one in flight); used for the rocprofv3 kernel-stats profile so kernels do not overlap")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-region", action="store_true", help="time the region once instead of three times (profiling runs)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launcher, rank set-up, process group (KH_BENCH_BACKEND=gloo), sharding, the timed loop's submit / wait / combine "
                    "logic and the line, with a stand-in shard whose every MSM 'result' is the SRS blinding base h (host code of the library); value is meaningless (CPU test tier)")
    ap.add_argument("--no-oplist", "--no-prover", dest="no_oplist", action="store_true", help="skip the ProverProof::create block")
    ap.add_argument("--no-pair", action="store_true", help="skip BASELINE config 5 (Pallas + Vesta pair) inside the prover block")
    ap.add_argument("--pair", action="store_true", help="BASELINE config 5 only: the Pallas + Vesta pair at 2^16 gates, then exit")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: become N ranks (one per GPU) under the torch.distributed launcher
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    if args.strong and args.log_n == LOG_N:
        args.log_n = 22
    total = (1 << args.log_n) if args.strong else world << args.log_n
    base["metric"] = "MSM Mscalar/s at 2^%d (%s)" % (args.log_n, args.curve.capitalize())
    wd = Watchdog(rank, world, base)
    wd.phase("process group", 300)

    import torch
    dist = None
    coll_dev = "cpu"
    backend = None
    t_pg = 0.0
    # KH_BENCH_FORCE_COLLECTIVE=1: form the process group and run the combine's collective even in a world of ONE -- how the 1-GPU box
    # exercises the RCCL call paths (torch's `nccl` all-gather of device tensors, and the in-library kh_comm_* one) that N > 1 uses
    force_coll = os.environ.get("KH_BENCH_FORCE_COLLECTIVE", "0") not in ("", "0")
    if world > 1 or force_coll:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
        # KH_BENCH_BACKEND=gloo lets several ranks share ONE GPU (functional check of the N>1 path on a single-GPU box);
        # the driver's multi-GPU runs use nccl (= RCCL over xGMI) with one GPU per rank
        backend = os.environ.get("KH_BENCH_BACKEND", "nccl")
        if not args.dry_run:
            ndev = max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank % ndev)
        if os.environ.get("KH_BENCH_FAIL_RANK") == str(rank):            # test hook (tests/test_bench_dry.py): this rank dies before the process group forms
            raise RuntimeError("KH_BENCH_FAIL_RANK: rank %d fails on purpose" % rank)
        t_pg0 = time.perf_counter()
        dist_mod.init_process_group(backend=backend, rank=rank, world_size=world)
        t_pg = time.perf_counter() - t_pg0
        dist = dist_mod
        coll_dev = "cuda" if backend == "nccl" else "cpu"

    import proof_systems_amd.khip as khip
    from proof_systems_amd import sharded
    if args.pair:
        khip.init(0)
        print(json.dumps({"metric": "Pallas + Vesta pair, ProverProof::create at 2^16 gates each", "unit": "constraints/s", **pair_block(khip, check_with_oracle=not args.no_cpu_baseline)}))
        return
    dev = local_rank % max(1, khip.device_count())
    cores = os.cpu_count() or 1

    # bases: this rank's point range of SRS::<curve>::create(total).g, generated and table-expanded on its own GPU
    t0 = time.perf_counter()
    CID = khip.VESTA if args.curve == "vesta" else khip.PALLAS
    # KH_BENCH_COMM=lib: the combine of the timed loop through the library's OWN collective (kh_comm_allgather_points + kh_points_sum = what
    # kh_msm_allreduce does and a Rust / C caller gets; librccl by dlopen, no torch in the data path) instead of torch.distributed's all_gather
    # Round 6: the in-library collective is the DEFAULT carrier of the timed loop whenever the process group is RCCL (`nccl`), torch's all_gather the
    # cross-check (other_collective_check) -- KH_BENCH_COMM=torch swaps the roles.  The communicator is formed under a time limit and the ranks then AGREE
    # (an all-reduce over the torch group) on whether every one of them has it: a rank that could not load librccl or join falls everybody back to torch.
    lib_comm, lib_note = None, None
    want_lib = os.environ.get("KH_BENCH_COMM", "lib" if backend == "nccl" else "torch") == "lib"
    affinity = {"pci": None, "numa_node": None, "cpus_bound": None}
    if not args.dry_run:
        affinity = bind_rank_to_gpu(torch, dev, world)
    if want_lib and dist is not None and not args.dry_run:
        wd.phase("in-library communicator", 240)
        khip.init(dev)
        box = {}

        def form():
            try:
                torch.cuda.set_device(dev)
                box["comm"] = make_lib_comm(khip, dist, rank, world)
            except Exception as e:                        # noqa: BLE001 -- reported, the torch collective takes over
                box["err"] = "%s: %s" % (type(e).__name__, e)
        th_ = threading.Thread(target=form, daemon=True); th_.start(); th_.join(120)
        ok_t = torch.tensor([1 if "comm" in box else 0], dtype=torch.int32, device=coll_dev)
        if th_.is_alive():
            lib_note = "kh_comm_init did not return within 120 s on rank %d" % rank       # (its thread holds the torch group inside a broadcast: nothing more can be agreed)
            raise RuntimeError(lib_note)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if int(ok_t.item()) == 1:
            lib_comm = box["comm"]
        else:
            lib_note = "in-library communicator not formed on every rank (%s): torch.distributed carries the loop" % box.get("err", "another rank failed")
            if "comm" in box:
                box["comm"].free()
    wd.phase("shard tables", 300)
    sm = sharded.RankShardedMsm(CID, total, dist=dist, coll_device=coll_dev, engine=_DryEngine(khip) if args.dry_run else sharded.KhipEngine(dev), rank=rank, world=world,
                                comm=lib_comm, always_collective=force_coll)
    srs, n = sm.shard, sm.count
    t_gen = time.perf_counter() - t0
    if world > 1 or force_coll:                            # one line per rank BEFORE the timed loop: what a first multi-GPU run needs to be debugged from its log
        print("[bench rank %d/%d] device %d (%d visible, pci %s, numa node %s, host threads bound to %s of its CPUs), shard = points [%d, %d) of %d, collective %s over %s, "
              "process group up in %.2f s, shard tables in %.2f s%s"
              % (rank, world, dev, khip.device_count(), affinity["pci"], affinity["numa_node"], affinity["cpus_bound"], sm.start, sm.start + n, total,
                 "rccl-lib" if lib_comm is not None else ("%s-torch" % backend), backend, t_pg, t_gen, ("; " + lib_note) if lib_note else ""), file=sys.stderr, flush=True)
    if args.dry_run:
        class _NoBuf:
            ptr = 0
        sc, d_sc = None, _NoBuf()
    else:
        sc = rand_scalars(np.random.default_rng(1234 + rank), n)
        d_sc = khip.DevBuf(sc.nbytes).upload(sc)

    depth = 1 if args.no_pipeline else int(os.environ.get('KH_BENCH_DEPTH', '2'))      # round 5, wide tables (profiles/r05_wide_sweep.txt): 2 / 3 / 4 in flight = 1007 / 998 / 1015 and 1009 / 1001 / 992 Mscalar/s: no difference; round 4: 852 / 836 / 821
    # one collective per MSM by default (batching `depth` partial sums into one collective made the N > 1 figure incomparable with
    # earlier rounds); KH_BENCH_COMBINE_EVERY=k batches k finished MSMs per collective as a prover would per phase
    combine_every = max(1, int(os.environ.get('KH_BENCH_COMBINE_EVERY', '1')))
    # KH_BENCH_RAMP=r: the number in flight starts at r and grows by one per finished MSM up to `depth` (an experiment on the pipeline's fill:
    # four jobs submitted at once run their sorts and accumulations in lockstep until they drift apart)
    ramp = int(os.environ.get('KH_BENCH_RAMP', '0'))
    cuda_dev = torch.cuda.current_device() if (coll_dev == "cuda" and not args.dry_run) else None

    def fence():
        if not args.dry_run:
            khip.sync()
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not args.dry_run:
                torch.cuda.synchronize()

    def measure(sm_, srs_, d_sc_, n_, nregions, label):
        """warm-up (synchronous steps: the latency), then `nregions` timed regions of EXACTLY args.steps MSMs each over the shard `srs_` of `sm_`; returns
        (region wall times -- max over ranks --, last combined result, last local partial, synchronous step times).  Timed region: `depth` MSMs in flight
        (kh_msm_submit / kh_msm_wait; two by default): the sort of step i+1 and the bucket-reduction tail of step i-1 run underneath the accumulation of
        step i.  Every step is a full MSM whose affine result is fetched and (N>1) combined across ranks INSIDE the region -- on the combiner thread, in
        submission order, so that the submitting thread is never inside a collective (round 6) --, one collective per MSM (combine_every)."""
        state = {"result": None, "partial": None}
        combiner = Combiner(sm_, torch, cuda_dev) if (dist is not None and (world > 1 or force_coll) and os.environ.get("KH_BENCH_INLINE_COMBINE", "0") in ("", "0")) else None

        def combine_now(out, inf):
            state["partial"] = (np.array(out[:1], dtype=np.uint64).reshape(1, 8), np.array(inf[:1], dtype=np.uint8).reshape(1))
            o, i = sm_.combine(out[:1], inf[:1])          # all-gather of the partial sums + local fold (no-op at N = 1)
            return o[0], bool(i[0])
        wd.phase(label + ": warm-up", 180)
        sync_ms_ = []
        for _ in range(max(1, args.warmup)):               # warm-up doubles as the latency measurement: synchronous steps
            ts = time.perf_counter()
            state["result"] = combine_now(*srs_.msm_batch_dev(d_sc_.ptr, n_, 1))
            sync_ms_.append(1e3 * (time.perf_counter() - ts))

        def timed_region():
            pending, done = [], []

            def flush():
                if not done:
                    return
                state["partial"] = (np.array(done[-1][0], dtype=np.uint64).reshape(1, 8), np.array([done[-1][1]], dtype=np.uint8))
                xs, fs = [d[0] for d in done], [d[1] for d in done]
                done.clear()
                if combiner is not None:
                    combiner.push(xs, fs)
                else:
                    o, i = sm_.combine(xs, fs)
                    state["result"] = (o[-1], bool(i[-1]))

            def collect(ticket):
                xy, inf = srs_.msm_wait(ticket)
                done.append((xy[0], inf[0]))
                if len(done) >= combine_every:
                    flush()
            fence()
            t0 = time.perf_counter()
            cur_depth = min(depth, ramp) if ramp > 0 else depth
            for _ in range(args.steps):
                pending.append(srs_.msm_submit(d_sc_.ptr, n_, 1))
                if len(pending) >= cur_depth:
                    collect(pending.pop(0))
                    cur_depth = min(depth, cur_depth + 1)
            while pending:
                collect(pending.pop(0))
            flush()
            if combiner is not None:
                state["result"] = combiner.drain()         # every combine of the region has landed before the clock stops
            fence()
            el = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el
        # The region runs THREE times back to back (single-region numbers moved 3-5 % box to box and run to run); `value` is the MEDIAN
        # region, all three are on the line (`value_runs`); ms_per_step x steps is that one region's wall time.
        if not args.dry_run:
            khip.set_phase_timers(False)                   # no per-phase HIP events inside the timed regions (the library's default; khip.init switches them on for tools)
        runs_ = []
        for r_ in range(nregions):
            wd.phase("%s: timed region %d" % (label, r_), 180)
            runs_.append(timed_region())
        if not args.dry_run:
            khip.set_phase_timers(True)                    # ... the synchronous steps below read them
        if combiner is not None:
            combiner.close()
        return runs_, state["result"], state["partial"], sync_ms_, combine_now

    nreg = 1 if args.single_region else 3
    runs, result, last_p, sync_ms, combine = measure(sm, srs, d_sc, n, nreg, "strong" if args.strong else "weak")
    last_partial = [last_p]
    elapsed = sorted(runs)[len(runs) // 2]

    # BOTH scalings from one invocation (round 6): at N > 1 the weak run above is followed by BASELINE config 4 -- ONE 2^22-point MSM whose point range is cut
    # over the ranks (2^22 / N points each) -- with its own shard tables, warm-up and timed regions; it rides in the line as `strong` (KH_BENCH_NO_STRONG=1
    # skips it; `--strong` alone still makes config 4 the line's own workload).
    strong_block = None
    if world > 1 and not args.strong and os.environ.get("KH_BENCH_NO_STRONG", "0") in ("", "0"):
        wd.phase("strong: shard tables", 300)
        s_total = 1 << int(os.environ.get("KH_BENCH_STRONG_LOG_N", "22"))
        sm2 = sharded.RankShardedMsm(CID, s_total, dist=dist, coll_device=coll_dev, engine=sm.engine, rank=rank, world=world, comm=lib_comm, always_collective=force_coll)
        if args.dry_run:
            sc2, d_sc2 = None, d_sc
        else:
            sc2 = rand_scalars(np.random.default_rng(4321 + rank), sm2.count)
            d_sc2 = khip.DevBuf(sc2.nbytes).upload(sc2)
        runs2, result2, _, sync2, _ = measure(sm2, sm2.shard, d_sc2, sm2.count, nreg, "strong")
        el2 = sorted(runs2)[len(runs2) // 2]
        strong_block = {"workload": "msm_2^%d_%s_srs_sharded" % (s_total.bit_length() - 1, args.curve), "scaling": "strong", "points_total": s_total, "points_per_gpu": sm2.count,
                        "value": None if args.dry_run else s_total / (el2 / args.steps) / 1e6, "unit": "Mscalar/s", "ms_per_step": 1e3 * el2 / args.steps,
                        "value_runs": [None if args.dry_run else s_total / (r / args.steps) / 1e6 for r in runs2], "ms_per_step_synchronous": float(np.median(sync2)),
                        "collective_backend": sm2.collective_backend}
        if args.dry_run:
            h = sm2.shard.h
            acc_xy, acc_inf = h.copy(), False
            for _ in range(world - 1):
                acc_xy, acc_inf = sm.engine.points_sum(CID, np.stack([acc_xy, h]), np.zeros(2, np.uint8))
            strong_block["combined_result_is_world_times_h"] = (not result2[1]) and bool(np.array_equal(np.asarray(result2[0], dtype=np.uint64).reshape(8), acc_xy))
        elif not args.no_cpu_baseline:
            wd.phase("strong: parity against the oracle", 600)
            from oracle import cref          # checker leg only
            want2, winf2 = cref.msm(CID, sm2.shard.get_g(), sc2, scalars_mont=True, threads=max(1, usable_threads() // world))
            mine = torch.from_numpy(np.concatenate([want2, np.array([int(winf2)], dtype=np.uint64)]).view(np.int64).copy()).to(coll_dev)
            allp = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allp, mine)
            parts = torch.stack(allp).cpu().numpy().view(np.uint64)
            accp, ainf = parts[0, :8].copy(), bool(parts[0, 8])
            for r in range(1, world):
                accp, ainf = cref.point_add(CID, accp, parts[r, :8].copy(), ainf, bool(parts[r, 8]))
            strong_block["combined_result_matches_oracle"] = bool((ainf == result2[1]) and (ainf or bool(np.array_equal(accp, result2[0]))))
        if not args.dry_run:
            d_sc2.free()
        sm2.close()
    wd.phase("after the timed regions", 900)

    ms_per_step = 1e3 * elapsed / args.steps
    value = total / (elapsed / args.steps) / 1e6
    if args.dry_run:                                       # the combined "result" must be world x h: the collective and the fold really ran
        h = srs.h
        acc_xy, acc_inf = h.copy(), False
        for _ in range(world - 1):
            acc_xy, acc_inf = sm.engine.points_sum(CID, np.stack([acc_xy, h]), np.zeros(2, np.uint8))
        ok = (not result[1]) and bool(np.array_equal(np.asarray(result[0], dtype=np.uint64).reshape(8), acc_xy))
        if rank == 0:
            print(json.dumps({"metric": "MSM Mscalar/s at 2^%d (%s)" % (args.log_n, args.curve.capitalize()), "dry_run": True, "value": None, "unit": "Mscalar/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "ranks_reported": world, "value_runs": [None] * len(runs),
                              "combined_result_is_world_times_h": ok, "strong": strong_block,
                              "combine_thread": bool(dist is not None and (world > 1 or force_coll) and os.environ.get("KH_BENCH_INLINE_COMBINE", "0") in ("", "0")),
                              "config": {"workload": "dry run: no GPU work", "collective_backend": sm.collective_backend, "process_group_backend": backend,
                                         "world_size_seen": world, "partials_per_collective": combine_every, "msm_in_flight": depth}}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    # Per-phase HIP events and the dominant kernel's own start/stop events (hipExtLaunchKernelGGL, on the stream the kernel
    # is launched on): synchronous steps right AFTER the timed loop, i.e. at the clocks the loop ran at, one MSM in flight
    # so that kernels do not overlap.  Median of the samples.  The same steps give the single-MSM latency.
    phase_ms, lat_ms = {}, []
    for _ in range(5):
        ts = time.perf_counter()
        result = combine(*srs.msm_batch_dev(d_sc.ptr, n, 1))
        lat_ms.append(1e3 * (time.perf_counter() - ts))
        for name, ms in khip.last_timings():
            phase_ms.setdefault(name, []).append(ms)
    phase_avg = {k: float(np.median(v)) for k, v in phase_ms.items()}
    kname = "k_acc_wide29" if "k_acc_wide29" in phase_avg else ("k_accumulate29" if "k_accumulate29" in phase_avg else "k_accumulate")
    acc = phase_avg.get(kname, phase_avg.get("accumulate"))
    latency = float(np.median(lat_ms))

    line = {
        "metric": "MSM Mscalar/s at 2^%d (%s)" % (args.log_n, args.curve.capitalize()), "value": value, "unit": "Mscalar/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u29x9 / u32x8 (255-bit Montgomery integers)",
        "data": "synthetic",
        "config": {"workload": ("msm_2^%d_%s_srs" % (args.log_n, args.curve)) + ("_sharded" if args.strong else ""), "points_per_gpu": n, "points_total": total,
                   "bases": "SRS::<%s>::create" % args.curve.capitalize(),
                   "scalars": "uniform 254-bit, seed 1234+rank", "parallelism": "point-range x%d" % world,
                   "collective_backend": sm.collective_backend, "process_group_backend": backend, "world_size_seen": world,
                   "partials_per_collective": (combine_every if sm.collective_backend else None)},
        "value_runs": [total / (r / args.steps) / 1e6 for r in runs], "value_note": "median of %d back-to-back timed regions of %d MSMs each; ms_per_step x steps = that region" % (len(runs), args.steps),
        "ranks_reported": world,
        "latency_value": total / (latency * 1e-3) / 1e6, "latency_note": "one MSM at a time (submit -> wait -> combine): `value` keeps %d in flight" % depth,
        "ms_per_step_synchronous": latency, "msm_in_flight": depth,
        "roofline": roofline_block(kname, acc, n, args.log_n) if not args.strong else roofline_block(kname, acc, n, -1),
        "phases_ms": phase_avg, "srs_create_device_s": t_gen,
        "combine_thread": bool(dist is not None and (world > 1 or force_coll) and os.environ.get("KH_BENCH_INLINE_COMBINE", "0") in ("", "0")),
        "rank0_locality": affinity,
    }
    if lib_note:
        line["collective_note"] = lib_note
    if strong_block is not None:
        line["strong"] = strong_block
    if world == 1 and not args.strong:
        wd.phase("host-scalar pipeline", 300)
        line.update(host_scalars_block(khip, srs, sc, n, args.steps, int(os.environ.get("KH_BENCH_HOST_DEPTH", "3")), result))     # 2 / 3 in flight: 798-849 / 901-933 Mscalar/s (profiles/r06_host_msm.txt)

    # parity at full size, every rank: its partial against the oracle on its own slice; rank 0 then folds the oracle's
    # partials with the oracle's group law and compares with the combined GPU result
    if not args.no_cpu_baseline:
        wd.phase("parity against the oracle / cpu baseline", 900)
        from oracle import cref          # cpu_baseline / checker leg only
        g = srs.get_g()
        threads = max(1, usable_threads() // world)
        t0 = time.perf_counter()
        want, winf = cref.msm(CID, g, sc, scalars_mont=True, threads=threads)
        t_cpu = time.perf_counter() - t0
        if world == 1:
            ok = (winf == result[1]) and (winf or bool(np.array_equal(want, result[0])))
            line["cpu_baseline"] = {"value": n / t_cpu / 1e6, "unit": "Mscalar/s", "cores": cref.last_threads(), "host_cores": cores, "cgroup_cpu_quota": cpu_quota(), "usable_cpus": len(os.sched_getaffinity(0)), "kind": "port",
                                    "sample": "the same 2^%d-point MSM (oracle/pasta_ref.c signed-window Pippenger; jobs = windows x point slices over the CPUs this container may use, window width chosen for that)" % args.log_n,
                                    "seconds": t_cpu, "gpu_result_matches": bool(ok)}
        else:
            mine = torch.from_numpy(np.concatenate([want, np.array([int(winf)], dtype=np.uint64)]).view(np.int64).copy()).to(coll_dev)
            allp = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allp, mine)
            ok = True
            if rank == 0:
                parts = torch.stack(allp).cpu().numpy().view(np.uint64)
                accp, ainf = parts[0, :8].copy(), bool(parts[0, 8])
                for r in range(1, world):
                    accp, ainf = cref.point_add(CID, accp, parts[r, :8].copy(), ainf, bool(parts[r, 8]))
                ok = (ainf == result[1]) and (ainf or bool(np.array_equal(accp, result[0])))
                line["multi_gpu_parity"] = {"combined_result_matches_oracle": bool(ok), "oracle_seconds_per_rank": t_cpu, "threads_per_rank": threads}
        if rank == 0 and not ok:
            line["parity_error"] = "GPU result differs from the CPU oracle"

    wd.phase("transform and prover blocks", 1800)
    if rank == 0 and world == 1 and not args.no_oplist and args.curve == "vesta" and not args.strong:
        line["ntt_kernels"] = ntt_block(khip)
        line["prover"] = prover_block(khip, srs, check_with_oracle=not args.no_cpu_baseline)
        if not args.no_pair:
            line["prover"]["pair"] = pair_block(khip, check_with_oracle=not args.no_cpu_baseline)

    # Both collectives in ONE run: whichever did not carry the timed loop repeats the last step's combine afterwards (never timed into `value`) and
    # must land on the same point -- the driver's N > 1 runs thereby execute the in-library RCCL path too.  Guarded by a watchdog: a collective
    # that hangs costs the check, not the line.
    hung = False
    wd.phase("the other collective", 240)
    if dist is not None and backend == "nccl" and last_partial[0] is not None:
        line["other_collective"], hung = other_collective_check(khip, dist, sm, lib_comm, last_partial[0], result, rank, world, coll_dev)
    wd.phase("teardown", 120)
    # LAST on the line (the driver's record keeps the tail of the output): both halves of BASELINE's metric and what pins them
    pr = line.get("prover") or {}
    line["summary"] = {"msm_Mscalar_per_s": value, "msm_Mscalar_per_s_host_scalars": line.get("value_host_scalars"), "n_gpus": world,
                       "prover_seconds": pr.get("seconds"), "prover_constraints_per_s": pr.get("constraints_per_s"), "prover_gpu_busy_frac": pr.get("gpu_busy_frac"),
                       "prover_byte_identical_to_oracle": pr.get("byte_identical_to_oracle"), "proofs_per_s_concurrent": (pr.get("concurrent") or {}).get("proofs_per_s"),
                       "msm_gpu_result_matches_oracle": (line.get("cpu_baseline") or {}).get("gpu_result_matches", (line.get("multi_gpu_parity") or {}).get("combined_result_matches_oracle")),
                       "cpu_baseline_cores": (line.get("cpu_baseline") or {}).get("cores"), "strong_Mscalar_per_s": (strong_block or {}).get("value")}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if hung:
        os._exit(0)                                   # a thread is stuck inside a collective: no orderly teardown is possible
    if lib_comm is not None:
        lib_comm.free()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
