#!/bin/bash
# The staged scatter of k_part2_sort at the other sizes of the wide path (2^19, 2^21, 2^22): direct (0) against the default, same box.
cd "$(dirname "$0")/.."
for ln in 19 21 22; do for s in 0 28672 0 28672; do
  KH_SWEEP_LOGN=$ln KH_PART2_STAGE=$s python tools/wide_sweep.py --child "2^$ln stage $s" 2>&1 | tail -1
done; done
