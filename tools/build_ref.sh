#!/bin/bash
# Builds the library of another commit beside the working tree's (for same-box A/Bs through KH_LIB): tools/build_ref.sh <git-ref> [out.so]
set -e
REF=${1:-HEAD}; ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$(realpath -m "${2:-$ROOT/proof_systems_amd/libkimchi_hip_B.so}")
T=$(mktemp -d)
git -C "$ROOT" archive "$REF" proof_systems_amd/csrc include | tar -x -C "$T"
cd "$T/proof_systems_amd/csrc"
for f in *.hip *.cpp; do
  if [[ $f == *.cpp ]]; then H="-mbmi2 -madx"; else H="-Xarch_host -mbmi2 -Xarch_host -madx"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $H -c $f -o ${f%.*}.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" *.o
rm -rf "$T"; ls -la "$OUT"
