#!/usr/bin/env python3
"""Collects the rocprofv3 evidence behind bench.py's `roofline` block, on the GPU box, for the build that is checked out.

Passes (each its own run, as MI355X_MICROARCH.md prescribes -- PMC counters are never combined with a trace):
  1. --kernel-trace                    bench.py --no-pipeline (one MSM in flight: kernels do not overlap) -> per-kernel stats
  2. --pmc FETCH_SIZE                  HBM read traffic per launch
  3. --pmc WRITE_SIZE                  HBM write traffic per launch
  4. --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
  5. --pmc GRBM_GUI_ACTIVE             busy cycles -> the clock the kernel actually sustained (DVFS)
Output: <out>.json = {"source_sha256": <hash of csrc/ at collection time>, "kernels": {name: {"calls", "avg_ns", counters...}}}
plus <out>_kernel_stats.csv.  bench.py refuses to quote traffic / instruction counts from a file whose source hash is
not the hash of the sources it is running (a stale PMC file is how round 1's roofline block went wrong).

Usage (from the repo root on the GPU box):  python tools/profile_msm.py gpurun_out/r02_msm20 [--workload msm|ntt|gates]"""
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "proof_systems_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cuh", ".hpp", ".inc", ".cpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def short(name):
    name = name.split("(")[0].replace("void ", "")
    return name.replace("kh::", "").replace("<kh::FqParams>", "<Fq>").replace("<kh::FpParams>", "<Fp>")


def run_pass(tag, rocprof_args, cmd, outdir):
    d = os.path.join(outdir, tag)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
    full = ["rocprofv3"] + rocprof_args + ["-d", d, "-o", tag, "--"] + cmd
    r = subprocess.run(full, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    open(os.path.join(outdir, tag + ".log"), "wb").write(r.stdout)
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if r.returncode != 0 or not dbs:
        print(f"pass {tag}: rc={r.returncode}, no database" if not dbs else f"pass {tag}: rc={r.returncode}")
        return None
    return dbs[0]


def kernel_stats(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    extra = ", avg(vgpr_count)" if "vgpr_count" in cols else ", null"
    rows = c.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start){extra} from kernels group by {namecol} order by 3 desc").fetchall()
    return rows


def counters(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    out = {}
    for k, cn, n, v in rows:
        out.setdefault(short(k), {})[cn] = v
        out[short(k)]["_launches_" + cn] = n
    return out


def main():
    out = sys.argv[1]
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "msm"
    outdir = out + "_passes"
    os.makedirs(outdir, exist_ok=True)
    if workload == "msm":
        cmd = [sys.executable, "bench.py", "--no-pipeline", "--single-region", "--no-cpu-baseline", "--no-oplist", "--steps", "10", "--warmup", "2"]
    elif workload == "gates":
        cmd = [sys.executable, "tools/gate_expr_time.py"]                   # every gate of the library on 2^19 rows: token machine and compiled kernel
    else:
        cmd = [sys.executable, "tools/bench_ntt.py", "--bench-shapes"]      # iNTT 2^16 x 19 + LDE 2^16 -> 2^19 x 16: what bench.py's ntt_kernels quotes
    res = {"source_sha256": source_hash(), "command": " ".join(cmd[1:]), "kernels": {}}
    db = run_pass("trace", ["--kernel-trace"], cmd, outdir)
    if db:
        lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs"]
        rows = kernel_stats(db)
        total = sum(r[2] for r in rows) or 1
        for name, calls, tot, avg, mn, mx, vg in rows:
            lines.append(f'"{name}",{calls},{tot},{avg:.1f},{mn},{mx},{100.0 * tot / total:.2f},{"" if vg is None else vg}')
            res["kernels"].setdefault(short(name), {}).update({"calls": calls, "avg_ns": avg, "min_ns": mn, "max_ns": mx})
        open(out + "_kernel_stats.csv", "w").write("\n".join(lines) + "\n")
    for tag, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                      ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"]),
                      ("grbm", ["GRBM_GUI_ACTIVE"])):
        db = run_pass(tag, ["--pmc"] + ctrs, cmd, outdir)
        if not db:
            continue
        for k, vals in counters(db).items():
            res["kernels"].setdefault(k, {}).update({cn: v for cn, v in vals.items() if not cn.startswith("_")})
    # units: FETCH_SIZE / WRITE_SIZE are in KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B: x2 for wide coalesced reads)
    for k, v in res["kernels"].items():
        if "FETCH_SIZE" in v:
            v["fetch_raw_bytes"] = v["FETCH_SIZE"] * 1024.0
        if "WRITE_SIZE" in v:
            v["write_bytes"] = v["WRITE_SIZE"] * 1024.0
        if "GRBM_GUI_ACTIVE" in v and v.get("avg_ns"):         # summed over the 8 XCDs; duration from the trace pass
            v["sustained_clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8.0 / v["avg_ns"]
    json.dump(res, open(out + "_pmc.json", "w"), indent=1)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("avg_ns", 0) * kv[1].get("calls", 0)):
        print(f"{k[:40]:40s} calls {v.get('calls', 0):4d} avg {v.get('avg_ns', 0) / 1e3:9.1f} us  fetch {v.get('fetch_raw_bytes', 0) / 1e6:9.2f} MB  write {v.get('write_bytes', 0) / 1e6:9.2f} MB  "
              f"VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:9.2f} M  GUI_ACTIVE {v.get('GRBM_GUI_ACTIVE', 0) / 1e6:7.2f} M")


if __name__ == "__main__":
    main()
