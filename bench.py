#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Kimchi hot path.

Workload at N=1 (BASELINE.json configs[1]): ONE 2^20-point Pippenger MSM over the Vesta SRS
(bases = SRS::<Vesta>::create(1<<20).g, generated on the device by kh_srs_create_device;
scalars = uniform 254-bit Fp Montgomery limbs from a fixed-seed PRNG), inputs resident in HBM
when the timed region starts.  A "step" is one such MSM through the C ABI (digits -> sort ->
bucket accumulation -> reduction -> host finish); the timed loop keeps four of them in flight
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
  oplist        (N=1) the op list of ProverProof::create at 2^16 gates (BASELINE
                config 3) replayed through the C ABI on synthetic columns: commitments, the opening
                rounds, NTTs and the vector steps between them -> constraints/s of the device side
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
ALG_BYTES_PER_PAIR = 96          # 64 B affine point + 32 B scalar, each read once
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_FILE = "r02_msm20_pmc.json"                 # tools/profile_msm.py (rocprofv3 PMC passes), keyed by the hash of csrc/
MIX_FILE = "r02_k_accumulate29_valu_mix.json"   # tools/valu_mix.py (static opcode histogram of the loop body)
RATES_FILE = "r02_valu_rates.json"              # per-opcode issue cycles measured by tools/microbench.hip


def rand_scalars(rng, n):
    s = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 62) - 1)          # < 2^254 < p: valid Montgomery limbs
    return s


def oplist_replay(khip, g16, srs16, reps=2):
    """SURVEY 3.1 totals at n = 2^16 (bench circuit): 15 Lagrange-basis commits (witness),
    1 + 7 monomial commits (z, t), the 16 opening rounds (L/R MSMs, a/b folds, challenge tensor; the basis
    fold is replaced by MSMs over the resident tables) + sg, 19 iNTT(n), 16 LDE(n->8n),
    iNTT(4n), iNTT(8n); plus the vector steps between them on the device (z accumulator, generic-gate and permutation
    constraint rows on d4 / d8, division by Z_H and the boundary quotients, chunked evaluations at two points, ft and
    opening input combinations).  Device-resident synthetic inputs; returns seconds per replay (best of reps)."""
    n = 1 << 16
    rng = np.random.default_rng(2024)
    F_R = np.array([0x34786d38fffffffd, 0x992c350be41914ad, 0xffffffffffffffff, 0x3fffffffffffffff], dtype=np.uint64)
    wit = np.tile(F_R, (15, n, 1))                       # 15 columns of Fp(1) (kimchi/src/bench.rs:106)
    wit[:, n - 10: n - 3] = 0
    wit[:, n - 3:] = rand_scalars(rng, 45).reshape(15, 3, 4)
    d_wit = khip.DevBuf(wit.nbytes).upload(wit)
    zt = rand_scalars(rng, 8 * n).reshape(8, n, 4)       # z and the 7 chunks of t: uniform stand-ins
    d_zt = khip.DevBuf(zt.nbytes).upload(zt)
    d_cols = khip.DevBuf(19 * n * 32).upload(rand_scalars(rng, 19 * n))
    d_lde_in = khip.DevBuf(16 * n * 32).upload(rand_scalars(rng, 16 * n))
    d_lde_out = khip.DevBuf(16 * 8 * n * 32)
    d_t4 = khip.DevBuf(4 * n * 32).upload(rand_scalars(rng, 4 * n))
    d_t8 = khip.DevBuf(8 * n * 32).upload(rand_scalars(rng, 8 * n))
    ipa_a = rand_scalars(rng, n); ipa_b = rand_scalars(rng, n); ipa_rand = rand_scalars(rng, 2)
    ipa_chals = [int.from_bytes(rng.bytes(16), "little") for _ in range(16)]
    u_base = khip.srs_generate(0, 1 << 21, 1)[0]
    # ---- the vector steps around the hot path (timing only: random columns; parity lives in tests/test_gpu_{expr,permutation,poly_ops}.py)
    import proof_systems_amd.polish as OP               # token streams of the generic gate / permutation constraints
    fid = khip.FP
    d1 = [khip.DevBuf(n * 32).upload(rand_scalars(rng, n)) for _ in range(15)]             # w0..6, sigma0..6, sid
    one = np.tile(F_R, (1, 1))
    num = khip.DevBuf(n * 32); den = khip.DevBuf(n * 32); ratio = khip.DevBuf(n * 32)
    k17 = rand_scalars(rng, 17)
    cell = OP.cell
    num_t, den_t = OP.perm_aggreg_tokens()
    d8cols = [khip.DevBuf(8 * n * 32).upload(rand_scalars(rng, 8 * n)) for _ in range(17)]
    gen_t = OP.generic_gate_tokens(0, 6, 16, 0, 1)
    perm_t = OP.perm_quot_tokens(w0=0, s0=7, z=14, x=15, zkpm=16, gamma=0, beta=1, bshift0=3, alpha0=2)
    t4 = khip.DevBuf(4 * n * 32); t8 = khip.DevBuf(8 * n * 32); tq = khip.DevBuf(7 * n * 32); tr = khip.DevBuf(n * 32); bq = khip.DevBuf(n * 32)
    polys = [khip.DevBuf(n * 32).upload(rand_scalars(rng, n)) for _ in range(44)] + [tq]      # ~45 polynomials enter the opening (prover.rs:1272-1477)
    plens = [n] * 44 + [7 * n]; pchunks = [1] * 44 + [7]
    pts2 = rand_scalars(rng, 2); sc45 = rand_scalars(rng, 45)
    a_dev = khip.DevBuf(n * 32); b_dev = khip.DevBuf(n * 32); ft = khip.DevBuf(n * 32)

    def vector_steps():
        khip.expr_evaluations_dev(fid, num_t, d1, [n] * 15, k17, n - 1, num, out_offset=1)        # perm_aggreg (permutation.rs:447-577)
        khip.expr_evaluations_dev(fid, den_t, d1, [n] * 15, k17, n - 1, den, out_offset=1)
        khip.batch_inversion_dev(fid, den, n - 1, offset=1)
        khip.expr_evaluations_dev(fid, [cell(0), cell(1), (OP.TOK_MUL, 0)], [num, den], [n, n], one, n, ratio)
        khip.field_scan_dev(fid, khip.SCAN_MUL, ratio, n - 2)
        khip.expr_evaluations_dev(fid, gen_t, d8cols, [8 * n] * 17, k17, 4 * n, t4, stride=2, next_shift=8)    # generic gate on d4 (prover.rs:806-812)
        khip.expr_evaluations_dev(fid, perm_t, d8cols, [8 * n] * 17, k17, 8 * n, t8, stride=1, next_shift=8)   # perm_quot on d8 (permutation.rs:237-283)
        khip.divide_by_linear_dev(fid, ratio, n, F_R, bq)                                         # bnd (permutation.rs:291-327)
        khip.divide_by_linear_dev(fid, ratio, n, k17[3], bq)
        khip.divide_by_vanishing_poly_dev(fid, t8, 8 * n, 16, tq, tr)                             # prover.rs:903
        khip.evaluate_chunks_batch_dev(fid, polys, plens, pchunks, n, pts2)                       # evaluations at zeta, zeta*omega (prover.rs:1028-1128)
        khip.poly_lincomb_dev(fid, polys[:20], [n] * 20, sc45[:20], ft, n)                        # ft (prover.rs:1147-1188)
        khip.combine_polys_dev(fid, polys, plens, pchunks, k17[0], n, a_dev)                      # SRS::open inputs (ipa.rs:852-888)
        khip.b_init_dev(fid, pts2, k17[1], n, b_dev)

    best = None
    phases = {}
    for _ in range(reps):
        khip.sync()
        t0 = time.perf_counter()
        ta = time.perf_counter()
        srs16.msm_batch_dev(d_wit.ptr, n, 15, basis=16)                       # witness commits
        srs16.msm_batch_dev(d_zt.ptr, n, 8)                                   # z + 7 t chunks
        tb = time.perf_counter()
        op = khip.IpaOpening(srs16, ipa_a, ipa_b, u_base)                     # SRS::open rounds (ipa.rs:929-1018)
        for ch in ipa_chals:
            op.round_lr(ipa_rand[0], ipa_rand[1])                             # the caller's sponge absorbs L, R here
            op.round_fold(ch)
        op.finish()
        op.free()
        tc = time.perf_counter()
        khip.ntt_dev(khip.FP, d_cols, 16, True, 19)
        khip.lde_dev(khip.FP, d_lde_in, 16, 3, d_lde_out, 16)
        khip.ntt_dev(khip.FP, d_t4, 18, True, 1)
        khip.ntt_dev(khip.FP, d_t8, 19, True, 1)
        khip.sync()
        td = time.perf_counter()
        vector_steps()
        khip.sync()
        t1 = time.perf_counter()
        if best is None or t1 - t0 < best:
            best = t1 - t0
            phases = {"commit_msm_s": tb - ta, "ipa_open_s": tc - tb, "ntt_s": td - tc, "vector_steps_s": t1 - td}
    # NTT kernels on their own (HIP events on the library stream): algorithmic bytes of SURVEY 8d
    def dev_ms(fn, reps=5):
        ts = []
        for _ in range(reps):
            fn(); khip.sync()
            ts.append(sum(ms for _, ms in khip.last_timings()))
        return float(np.median(ts))
    t_intt = dev_ms(lambda: khip.ntt_dev(khip.FP, d_cols, 16, True, 19))
    t_lde = dev_ms(lambda: khip.lde_dev(khip.FP, d_lde_in, 16, 3, d_lde_out, 16))
    phases["ntt_kernels"] = {
        "intt_2^16_x19": {"ms": t_intt, "algorithmic_GBps": 64.0 * n * 19 / (t_intt * 1e-3) / 1e9, "hbm_frac": 64.0 * n * 19 / (t_intt * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "lde_2^16_to_2^19_x16": {"ms": t_lde, "algorithmic_GBps": 288.0 * n * 16 / (t_lde * 1e-3) / 1e9, "hbm_frac": 288.0 * n * 16 / (t_lde * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "note": "VALU-issue bound like the MSM (about log2(N)/2 + 2 Montgomery products per element), see DESIGN.md section 4"}
    for b in [d_wit, d_zt, d_cols, d_lde_in, d_lde_out, d_t4, d_t8, num, den, ratio, t4, t8, tq, tr, bq, a_dev, b_dev, ft] + d1 + d8cols + polys[:44]:
        b.free()
    return best, phases


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
        clk = k.get("sustained_clock_ghz")
        block["valu_issue"] = {"instructions_per_launch": k["SQ_INSTS_VALU"], "issue_cycles_per_instruction": per_instr,
                               "issue_bound_ms_at_2.4GHz": peak_ms, "frac": peak_ms / acc_ms,
                               "sustained_clock_ghz": clk, "frac_at_sustained_clock": (peak_ms * 2.4 / clk / acc_ms) if clk else None,
                               "unit": "fraction of the issue-bound time for this instruction mix (per-opcode cycles from tools/microbench.hip)"}
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="weak scaling: points PER GPU; with --strong: points in total")
    ap.add_argument("--strong", action="store_true", help="BASELINE config 4: ONE MSM of 2^log-n points (default 2^22) sharded over the ranks")
    ap.add_argument("--curve", choices=["vesta", "pallas"], default="vesta")
    ap.add_argument("--no-pipeline", action="store_true", help="time synchronous MSMs (one in flight); used for the rocprofv3 kernel-stats profile so kernels do not overlap")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oplist", action="store_true")
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

    import torch
    dist = None
    coll_dev = "cpu"
    backend = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # KH_BENCH_BACKEND=gloo lets several ranks share ONE GPU (functional check of the N>1 path on a single-GPU box);
        # the driver's multi-GPU runs use nccl (= RCCL over xGMI) with one GPU per rank
        backend = os.environ.get("KH_BENCH_BACKEND", "nccl")
        ndev = max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank % ndev)
        dist_mod.init_process_group(backend=backend, rank=rank, world_size=world)
        dist = dist_mod
        coll_dev = "cuda" if backend == "nccl" else "cpu"

    import proof_systems_amd.khip as khip
    from proof_systems_amd import sharded
    dev = local_rank % max(1, khip.device_count())
    cores = os.cpu_count() or 1

    # bases: this rank's point range of SRS::<curve>::create(total).g, generated and table-expanded on its own GPU
    t0 = time.perf_counter()
    CID = khip.VESTA if args.curve == "vesta" else khip.PALLAS
    sm = sharded.RankShardedMsm(CID, total, dist=dist, coll_device=coll_dev, engine=sharded.KhipEngine(dev), rank=rank, world=world)
    srs, n = sm.shard, sm.count
    t_gen = time.perf_counter() - t0
    sc = rand_scalars(np.random.default_rng(1234 + rank), n)
    d_sc = khip.DevBuf(sc.nbytes).upload(sc)

    def combine(out, inf):
        o, i = sm.combine(out[:1], inf[:1])          # all-gather of the partial sums + local fold (no-op at N = 1)
        return o[0], bool(i[0])

    def fence():
        khip.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up doubles as the latency measurement: synchronous steps
    sync_ms = []
    for _ in range(max(1, args.warmup)):
        ts = time.perf_counter()
        result = combine(*srs.msm_batch_dev(d_sc.ptr, n, 1))
        sync_ms.append(1e3 * (time.perf_counter() - ts))
    # timed region: EXACTLY `steps` MSMs, four in flight (kh_msm_submit / kh_msm_wait): the sort of step i+2 and the
    # bucket-reduction tail of step i run underneath the accumulation of step i+1.  Every step is a full MSM whose affine
    # result is fetched and (N>1) combined across ranks.
    fence()
    t0 = time.perf_counter()
    depth = 1 if args.no_pipeline else 4
    pending = []
    for _ in range(args.steps):
        pending.append(srs.msm_submit(d_sc.ptr, n, 1))
        if len(pending) >= depth:
            result = combine(*srs.msm_wait(pending.pop(0)))
    while pending:
        result = combine(*srs.msm_wait(pending.pop(0)))
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    value = total / (elapsed / args.steps) / 1e6
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
    kname = "k_accumulate29" if "k_accumulate29" in phase_avg else "k_accumulate"
    acc = phase_avg.get(kname, phase_avg.get("accumulate"))
    latency = float(np.median(lat_ms))

    line = {
        "metric": "MSM Mscalar/s at 2^%d (%s)" % (args.log_n, args.curve.capitalize()), "value": value, "unit": "Mscalar/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u29x9 / u32x8 (255-bit Montgomery integers)",
        "data": "synthetic",
        "config": {"workload": ("msm_2^%d_%s_srs" % (args.log_n, args.curve)) + ("_sharded" if args.strong else ""), "points_per_gpu": n, "points_total": total,
                   "bases": "SRS::<%s>::create" % args.curve.capitalize(),
                   "scalars": "uniform 254-bit, seed 1234+rank", "parallelism": "point-range x%d" % world, "collective_backend": backend, "world_size_seen": world},
        "latency_value": total / (latency * 1e-3) / 1e6, "latency_note": "one MSM at a time (submit -> wait -> combine): `value` keeps 4 in flight",
        "ms_per_step_synchronous": latency, "msm_in_flight": depth,
        "roofline": roofline_block(kname, acc, n, args.log_n) if not args.strong else roofline_block(kname, acc, n, -1),
        "phases_ms": phase_avg, "srs_create_device_s": t_gen,
    }

    # parity at full size, every rank: its partial against the oracle on its own slice; rank 0 then folds the oracle's
    # partials with the oracle's group law and compares with the combined GPU result
    if not args.no_cpu_baseline:
        from oracle import cref          # cpu_baseline / checker leg only
        g = srs.get_g()
        threads = max(1, cores // world)
        t0 = time.perf_counter()
        want, winf = cref.msm(CID, g, sc, scalars_mont=True, threads=threads)
        t_cpu = time.perf_counter() - t0
        if world == 1:
            ok = (winf == result[1]) and (winf or bool(np.array_equal(want, result[0])))
            line["cpu_baseline"] = {"value": n / t_cpu / 1e6, "unit": "Mscalar/s", "cores": cref.last_threads(), "host_cores": cores, "kind": "port",
                                    "sample": "the same 2^%d-point MSM (oracle/pasta_ref.c signed-window Pippenger, threads = windows x point slices)" % args.log_n,
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

    if rank == 0 and world == 1 and not args.no_oplist and args.curve == "vesta" and not args.strong:
        g16 = srs.get_g(0, 1 << 16)
        srs16 = khip.Srs(khip.VESTA, g16)
        t0 = time.perf_counter()
        srs16.compute_lagrange(16)            # SRS::lagrange_basis as a device group-iNTT (index time, ipa.rs:1065-1172)
        t_lag = time.perf_counter() - t0
        t_op, ph = oplist_replay(khip, g16, srs16)
        line["oplist"] = {"workload": "ProverProof::create op list at 2^16 gates (23 MSM(n) + the 16 opening rounds of SRS::open incl. folds and sg + 19 iNTT(n) + 16 LDE(n->8n) + iNTT(4n) + iNTT(8n))",
                          "seconds": t_op, "constraints_per_s": (1 << 16) / t_op, **ph, "lagrange_basis_index_time_s": t_lag,
                          "note": "MSM + NTT + opening rounds + the vector steps between them (z accumulator, generic-gate and permutation rows, divisions, chunked evaluations, combinations) on synthetic columns; sponge, RNG and the closed-form Lagrange evaluations stay on the host (SURVEY 8d cfg 3)"}

    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
