#!/usr/bin/env python3
"""profiles/<round>_microbench.txt (tools/microbench.hip) -> profiles/<round>_valu_rates.json: issue cycles per wave64
instruction per SIMD at 8 waves/SIMD (2.4 GHz), keyed by mnemonic; `default_cycles` prices opcodes that were not measured
(4.2: the slow class, conservative for the roofline fraction).  Usage: tools/parse_microbench.py in.txt out.json"""
import json
import re
import sys

cyc = {}
for ln in open(sys.argv[1]):
    m = re.match(r"\s*(v_[a-z0-9_]+).*waves/SIMD=8\s+[\d.]+ ms\s+[\d.]+ G wave-instr/s\s+([\d.]+) cycles", ln)
    if m and m.group(1) not in cyc:
        cyc[m.group(1)] = float(m.group(2))
# A measurement above 8 cycles is not an issue rate: the v_cndmask_b32 loop of tools/microbench.hip (every instruction reads VCC, which
# nothing in the loop writes) reports 23.5 cycles, yet k_ntt_pass -- 5.5 % of whose VALU instructions are v_cndmask -- would then need
# 259 us for the 105.5 M instructions it retires in 218 us (profiles/r03_ntt_pmc.json).  Such entries are dropped (-> default_cycles,
# the slow class) and listed under "rejected".
rejected = {k: v for k, v in cyc.items() if v > 8.0}
cyc = {k: v for k, v in cyc.items() if v <= 8.0}
alias = {"v_addc_co_u32": ["v_subb_co_u32", "v_subbrev_co_u32", "v_add_co_u32", "v_subrev_co_u32"], "v_sub_u32": ["v_subrev_u32"], "v_cmp_le_u32": ["v_cmp_lt_u32", "v_cmp_ge_u32", "v_cmp_eq_u32", "v_cmp_ne_u32", "v_cmp_gt_u32"],
         "v_mov_b32": ["v_not_b32"], "v_alignbit_b32": ["v_perm_b32"], "v_sub_co_u32": []}
for k, vs in alias.items():
    for v in vs:
        if k in cyc:
            cyc.setdefault(v, cyc[k])
json.dump({"source": sys.argv[1], "unit": "cycles per wave64 instruction per SIMD at 8 waves/SIMD, 2.4 GHz", "default_cycles": 4.2, "cycles": cyc, "rejected": rejected}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(cyc, indent=1))
