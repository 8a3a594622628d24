#!/bin/bash
# Same-box A/B of two builds of the library: A = proof_systems_amd/libkimchi_hip.so, B = $1 (KH_LIB).  Alternates the prover's latency and the MSM's pipelined rate.
B=${1:-/root/repo/proof_systems_amd/libkimchi_hip_B.so}
ROUNDS=${2:-3}
cd "$(dirname "$0")/.."
for r in $(seq $ROUNDS); do
  for v in A B; do
    if [ $v = B ]; then export KH_LIB=$B; else unset KH_LIB; fi
    echo "== $v round $r"
    [ -z "$AB_NO_PROVER" ] && python tools/prover_time.py 16 --native 2>&1 | grep "native check=False"
    python tools/wide_sweep.py --child "$v wide" 2>&1 | tail -1
    [ -z "$AB_NO_NARROW" ] && KH_WIDE_MIN_N=0 python tools/wide_sweep.py --child "$v narrow" 2>&1 | tail -1
  done
done
