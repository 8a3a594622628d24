B=$PWD/proof_systems_amd/libkimchi_hip_B.so
for r in 1 2; do
for w in 4 2 8 0; do echo "== A waves/CU $w"; KH_IPA_REBASE_WAVES_PER_CU=$w python tools/ipa_time.py 2>&1 | grep "per round\|rep 2" | tail -2 | cut -c1-330; KH_IPA_REBASE_WAVES_PER_CU=$w python tools/prover_time.py 16 --native 2>&1 | grep "native check=False\|folded basis" | cut -c1-220; done
echo "== B"; KH_LIB=$B python tools/ipa_time.py 2>&1 | grep "per round\|rep 2" | tail -2 | cut -c1-330; KH_LIB=$B python tools/prover_time.py 16 --native 2>&1 | grep "native check=False\|folded basis" | cut -c1-220
done
