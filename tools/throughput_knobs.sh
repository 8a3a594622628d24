for r in 1 2; do
for e in "KH_A=1" "KH_QUAD=0" "KH_NO_BSUM_QUAD=1" "KH_QUAD=0 KH_NO_BSUM_QUAD=1 KH_NO_FIN_QUAD=1" "KH_QUAD=0 KH_NO_BSUM_QUAD=1 KH_NO_FIN_QUAD=1 KH_NO_FUSED_SORT=1" "KH_QUAD_MAXG=0"; do
  for t in 1 6; do echo "== [$e] provers $t: $(env $e python tools/concurrent_provers.py $t 20 2>&1 | tail -1 | cut -c1-60)"; done
done; done
