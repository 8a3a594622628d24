#!/bin/bash
# Same-box sweep of k_part2_sort's staging size (KH_PART2_STAGE entries per pass; 0 = the direct scatter of round 5): synchronous phases + pipelined rate, two rounds.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for s in ${STAGES:-0 14336 20480 24576 28672}; do
  KH_PART2_STAGE=$s python tools/wide_sweep.py --child "stage $s" 2>&1 | tail -1
done; done
