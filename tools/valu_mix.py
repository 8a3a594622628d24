#!/usr/bin/env python3
"""Static instruction mix of the accumulation kernel's inner loop, for the VALU-issue roofline of bench.py.

VALU instructions do not all cost the same on gfx950 (tools/microbench.hip, profiles/r02_microbench.txt): v_mov / v_add_u32 /
v_sub_u32 / v_and / v_or / shifts / v_fma_f32 issue in ~2.4 cycles per wave64 instruction per SIMD, everything else the
kernel uses (v_mad_u64_u32, v_bfi, v_alignbit, v_lshrrev_b64, carry ops, v_add3) in ~4.2-4.4.  The issue-bound time of a
launch is therefore  sum over opcodes (executed count x measured cycles), with the executed count = SQ_INSTS_VALU (PMC)
split by the opcode histogram of the loop body's dominant basic block (the mixed addition: ~95 % of the loop).

Compiles csrc/msm.hip to gfx950 assembly (hipcc cross-compiles without a GPU) and writes
profiles/<round>_k_accumulate29_valu_mix.json: {"source_sha256", "kernel", "block_instructions", "valu_histogram"}.
Usage: tools/valu_mix.py profiles/r02_k_accumulate29_valu_mix.json"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from profile_msm import source_hash  # noqa: E402

KERNEL = "_ZN2kh12k_acc_wide29INS_8FqParamsE"           # the accumulation of the wide-window path (round 5); --kernel narrow: k_accumulate29
NARROW_KERNEL = "_ZN2kh14k_accumulate29INS_8FqParamsE"
NTT_KERNEL = "_ZN2kh10k_ntt_passINS_8FpParamsELi256E"       # tools/valu_mix.py OUT --kernel ntt: the whole kernel (straight-line radix-4 steps)


def main():
    out = sys.argv[1]
    ntt = "--kernel" in sys.argv and sys.argv[sys.argv.index("--kernel") + 1] == "ntt"
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm,
                               os.path.join(ROOT, "proof_systems_amd", "csrc", "ntt.hip" if ntt else "msm.hip")], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    if ntt:
        start = next(i for i, l in enumerate(lines) if l.startswith(NTT_KERNEL) and l.split(";")[0].strip().endswith(":"))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        ops = [l.split(";")[0].strip().split()[0] for l in lines[start + 1:end] if l.split(";")[0].strip() and not l.split(";")[0].strip().startswith(".") and not l.split(";")[0].strip().endswith(":")]
        hist = Counter(op for op in ops if op.startswith("v_"))
        res = {"source_sha256": source_hash(), "kernel": "k_ntt_pass<FpParams, 256>", "block": "whole kernel (static)", "block_instructions": len(ops),
               "block_valu_instructions": sum(hist.values()), "valu_histogram": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}
        json.dump(res, open(out, "w"), indent=1)
        print(json.dumps(res, indent=1)[:600])
        return
    narrow = "--kernel" in sys.argv and sys.argv[sys.argv.index("--kernel") + 1] == "narrow"
    kern = NARROW_KERNEL if narrow else KERNEL
    start = next(i for i, l in enumerate(lines) if l.startswith(kern) and l.split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = {"entry": []}, "entry"
    for l in lines[start + 1:end]:
        l = l.split(";")[0].strip()
        if not l:
            continue
        if l.endswith(":") and l.startswith(".LBB"):
            cur = l[:-1]; blocks[cur] = []
        elif not l.startswith("."):
            blocks[cur].append(l.split()[0])
    name, body = max(blocks.items(), key=lambda kv: sum(op == "v_mad_u64_u32" for op in kv[1]))
    hist = Counter(op for op in body if op.startswith("v_"))
    res = {"source_sha256": source_hash(), "kernel": ("k_accumulate29" if narrow else "k_acc_wide29") + "<FqParams>", "block": name, "block_instructions": len(body),
           "block_valu_instructions": sum(hist.values()), "valu_histogram": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
