#!/usr/bin/env python3
"""Sweep of the wide-window MSM knobs (csrc/msm.hip: KH_WIDE_OG, KH_WIDE_ACC_BLOCKS, KH_WIDE_RLOG are read once per process, so every configuration
runs in its own child process): synchronous per-phase times and the pipelined rate at depth 2 / 3 / 4 of the 2^20-point Vesta MSM.
Usage: wide_sweep.py            (parent: runs the list below)       wide_sweep.py --child TAG"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = [("default (og16, 3 blk/CU, r16)", {}), ("og4", {"KH_WIDE_OG": "4"}), ("og64", {"KH_WIDE_OG": "64"}), ("4 blk/CU", {"KH_WIDE_ACC_BLOCKS": "4"}),
           ("2 blk/CU", {"KH_WIDE_ACC_BLOCKS": "2"}), ("r8", {"KH_WIDE_RLOG": "3"}), ("r32", {"KH_WIDE_RLOG": "5"}), ("narrow c=16", {"KH_WIDE_MIN_N": "0"}),
           ("default again", {})]


def child(tag):
    import proof_systems_amd.khip as khip
    n = 1 << int(os.environ.get("KH_SWEEP_LOGN", "20"))
    khip.init(0)
    srs = khip.Srs.create(khip.VESTA, n)
    sc = np.random.default_rng(1).integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 61) - 1)
    d = khip.DevBuf(sc.nbytes).upload(sc)

    def pipelined(depth, steps=30):
        khip.sync(); t0 = time.perf_counter(); pend = []
        for _ in range(steps):
            pend.append(srs.msm_submit(d.ptr, n, 1))
            if len(pend) >= depth:
                srs.msm_wait(pend.pop(0))
        while pend:
            srs.msm_wait(pend.pop(0))
        khip.sync()
        return n / ((time.perf_counter() - t0) / steps) / 1e6
    acc = {}
    for _ in range(3):
        srs.msm_batch_dev(d.ptr, n, 1)
    for _ in range(7):
        srs.msm_batch_dev(d.ptr, n, 1)
        for k, v in khip.last_timings():
            acc.setdefault(k, []).append(v)
    ph = {k: np.median(v) * 1e3 for k, v in acc.items()}
    tot = sum(v for k, v in ph.items() if not k.startswith("k_"))
    r = []
    if not os.environ.get("KH_SWEEP_TIMERS_ON"):
        khip.set_phase_timers(False)                               # as bench.py's timed regions
    for depth in (2, 3, 4):
        pipelined(depth, 10)
        r.append(sorted(pipelined(depth) for _ in range(3))[1])
    kacc = ph.get("k_acc_wide29", ph.get("k_accumulate29", 0))
    print(f"{tag:30s} sync {tot:5.0f} us (acc kernel {kacc:4.0f}, a1 {ph.get('reduce_a1', 0):3.0f}, rest of reduce {ph.get('reduce', 0):3.0f}, sort {ph.get('scatter', 0):3.0f}) | "
          f"pipelined depth 2 / 3 / 4: " + " / ".join(f"{x:.0f}" for x in r) + " Mscalar/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for tag, env in CONFIGS:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag], env=dict(os.environ, **env))
