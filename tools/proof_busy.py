#!/usr/bin/env python3
"""GPU-busy fraction of ONE 2^16 proof through kh_prove, from a rocprofv3 kernel-trace database of `tools/prover_time.py 16 --native` (the last proof of
the run): wall time from the proof's first kernel to its last, the time with at least one kernel running (union of the kernel intervals over all streams),
the same two figures for the stretch before the first opening round, and the idle time split into dispatch-sized gaps (< 12 us: a dependent launch the host
had not queued ahead) and host-sized gaps (the transcript's Poseidon absorbs, the MSM's host finish).  Writes JSON (with the hash of csrc/, like the PMC
files: bench.py quotes it as prover.gpu_busy_frac only on the build it was measured on).  Usage: proof_busy.py results.db out.json"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "proof_systems_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cuh", ".hpp", ".inc", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def union_busy(rows):
    busy, end_so_far, gaps = 0, None, []
    for _, s, e in rows:
        if end_so_far is None:
            end_so_far = s
        if s > end_so_far:
            gaps.append((s - end_so_far) / 1e3)
        if e > end_so_far:
            busy += e - max(s, end_so_far); end_so_far = e
    return busy / 1e3, gaps


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    steps = [i for i, r in enumerate(rows) if "k_ipa_step" in r[0]]
    rounds = 16
    assert len(steps) >= 2 * rounds, "need at least two proofs in the trace"
    first_step = steps[-rounds]
    # the last proof starts after the largest idle gap between the previous proof's last round and this proof's first round
    seg = rows[steps[-rounds - 1]:first_step]
    ends = []
    m = 0
    for r in seg:
        m = max(m, r[2]); ends.append(m)
    gaps = [(seg[i + 1][1] - ends[i], i) for i in range(len(seg) - 1)]
    start_i = steps[-rounds - 1] + 1 + max(gaps)[1]
    proof = rows[start_i:]
    t0, t1 = proof[0][1], max(r[2] for r in proof)
    busy, idle_gaps = union_busy(proof)
    pre = rows[start_i:first_step]
    pre_busy, pre_gaps = union_busy(pre)
    pre_wall = (rows[first_step][1] - t0) / 1e3
    # the same stretch counted from the witness commitment's first kernel: since round 6 the proof's first kernels (the first column groups' transforms) run
    # UNDER the witness transfer, so the stretch above contains the transfer's ~0.65 ms; the commitment is queued behind the last group
    dig = next((i for i in range(start_i, first_step) if "k_digits" in rows[i][0]), start_i)
    post = rows[dig:first_step]
    post_busy, post_gaps = union_busy(post)
    post_wall = (rows[first_step][1] - rows[dig][1]) / 1e3
    out = {"source_sha256": source_hash(), "workload": "last proof of tools/prover_time.py 16 --native under rocprofv3 --kernel-trace",
           "kernels": len(proof), "wall_us": (t1 - t0) / 1e3, "busy_us": busy, "gpu_busy_frac": busy / ((t1 - t0) / 1e3),
           "pre_opening": {"wall_us": pre_wall, "busy_us": pre_busy, "idle_us": pre_wall - pre_busy, "kernels": len(pre),
                           "idle_in_gaps_under_12us": sum(g for g in pre_gaps if g < 12), "gaps_under_12us": sum(1 for g in pre_gaps if g < 12),
                           "idle_in_gaps_over_12us": sum(g for g in pre_gaps if g >= 12), "gaps_over_12us": sum(1 for g in pre_gaps if g >= 12)},
           "pre_opening_after_transfer": {"wall_us": post_wall, "busy_us": post_busy, "idle_us": post_wall - post_busy, "kernels": len(post),
                                          "idle_in_gaps_under_12us": sum(g for g in post_gaps if g < 12), "gaps_under_12us": sum(1 for g in post_gaps if g < 12),
                                          "idle_in_gaps_over_12us": sum(g for g in post_gaps if g >= 12), "gaps_over_12us": sum(1 for g in post_gaps if g >= 12),
                                          "note": "from the witness commitment's first kernel (queued behind the last column group's transfer) to the first opening round; gaps under 12 us "
                                                  "are dispatch-sized -- under the tracer every plainly launched kernel starts ~5 us late (tools/latency/chain_gap: 1.0-1.6 us "
                                                  "without it) --, gaps over 12 us are the host's (transcript absorbs, the batch inversion's turn, the evaluations' download)"},
           "opening": {"wall_us": (t1 - rows[first_step][1]) / 1e3, "busy_us": busy - pre_busy},
           "note": "wall = first kernel start to last kernel end of the proof (the witness upload in front and the host's last few us behind are outside); under the "
                   "profiler every launch costs the host a little more than in a plain run, so the fraction is a lower bound"}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
