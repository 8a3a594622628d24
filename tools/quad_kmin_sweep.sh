grep -m1 "model name" /proc/cpuinfo; grep -c processor /proc/cpuinfo; grep -o -m1 "avx512ifma" /proc/cpuinfo | head -1; grep -o -m1 "avx512f" /proc/cpuinfo | head -1
for q in 256 512 1024; do for km in 8 16; do echo "== KH_QUAD=$q KH_KMIN=$km"; KH_QUAD=$q KH_KMIN=$km python tools/prover_time.py 16 2>&1 | grep "check=False"; done; done
echo "== KH_KMIN=4"; KH_KMIN=4 python tools/prover_time.py 16 2>&1 | grep "check=False"
