for rep in 1 2; do
for e in "KH_IPA_REBASE=0" "KH_IPA_REBASE_LOGN=10" "KH_IPA_REBASE_LOGN=11" "KH_IPA_REBASE_LOGN=12" "KH_IPA_REBASE_LOGN=10 KH_IPA_REBASE_CUS=64" "KH_IPA_REBASE_LOGN=11 KH_IPA_REBASE_CUS=64" "KH_IPA_REBASE_LOGN=11 KH_IPA_REBASE_CUS=128" "KH_IPA_REBASE_LOGN=9"; do
  echo "== $e"; env $e python tools/prover_time.py 16 --native 2>&1 | grep "native check=False" | cut -c1-200
done; done
