#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
in separate runs, as MI355X_MICROARCH.md prescribes).  gfx950 correction from that guide:
FETCH_SIZE under-reports wide coalesced reads by exactly 2x (128-B requests tallied at 64 B);
our own calibration on known byte counts in THIS workload: k_digits reads 2^20 x 32 B = 32 MiB
and FETCH_SIZE reports 16.0 MiB; k_digits writes 64 MiB and WRITE_SIZE reports 64.0 MiB.
Usage: tools/pmc_traffic.py fetch.db write.db out.json"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_units": "bytes per launch", "_fetch_correction": 2.0, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, (0, 0.0))[1] * 1024.0
    w = write.get(k, (0, 0.0))[1] * 1024.0
    short = k.split("(")[0].replace("void ", "")
    out["kernels"][short] = {"launches": fetch.get(k, (0, 0))[0], "fetch_raw_bytes": f, "fetch_corrected_bytes": 2.0 * f,
                             "write_bytes": w, "hbm_bytes": 2.0 * f + w}
acc = [v for k, v in out["kernels"].items() if "k_accumulate" in k]
if acc:
    # k_accumulate's reads are 64-byte gathers (one affine point per lane), not a wide coalesced
    # stream: calibrated on its known byte count (16 windows x 2^20 points x 64 B + 67 MB of
    # entries = 1141 MB) the RAW counter reads 1401 MB, i.e. the 2x under-count does NOT apply
    # to this access pattern (it slightly over-counts: 128-B line fills for 64-B needs).
    out["k_accumulate_hbm_bytes_per_launch"] = acc[0]["fetch_raw_bytes"] + acc[0]["write_bytes"]
    out["k_accumulate_note"] = "raw FETCH_SIZE + WRITE_SIZE; 2x correction not applied (64-B gathers, calibrated)"
try:                                    # keep the SQ_INSTS_VALU figures of an earlier, separate pass
    old = json.load(open(sys.argv[3]))
    for k in ("k_accumulate_valu_wave_instructions_per_launch", "k_accumulate_sq_note"):
        if k in old:
            out[k] = old[k]
except (OSError, ValueError):
    pass
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k[:48]:48s} fetch(raw) {v['fetch_raw_bytes']/1e6:10.2f} MB  write {v['write_bytes']/1e6:10.2f} MB")
