for c in 13 16 15 14 12 10 9; do for g in 1 0; do
  r=$(KH_IPA_REBASE_C=$c KH_IPA_REBASE_GLV=$g KH_IPA_REBASE_WAIT=1 timeout 200 python -m pytest tests/test_gpu_proof_fixtures.py -x -q -k "test_kh_prove_reproduces_the_committed_proof_bytes and bench_vesta_2_16" 2>&1 | grep -E "passed|failed|first in" | tr "\n" " " | cut -c1-160)
  echo "c=$c glv=$g: $r"
done; done
