#!/bin/bash
# After `gpurun -- 'tools/microbench > gpurun_out/r02_microbench.txt; python tools/profile_msm.py gpurun_out/r02_msm20; python bench.py > gpurun_out/r02_bench_n1.json'`
# copy the evidence bench.py's roofline block quotes into profiles/ (tracked) and derive the two small JSON files from it.
set -e
cd "$(dirname "$0")/.."
R=${1:-r03}
cp gpurun_out/${R}_microbench.txt profiles/${R}_microbench.txt
cp gpurun_out/${R}_msm20_pmc.json profiles/${R}_msm20_pmc.json
cp gpurun_out/${R}_msm20_kernel_stats.csv profiles/${R}_msm20_kernel_stats.csv
python3 tools/parse_microbench.py profiles/${R}_microbench.txt profiles/${R}_valu_rates.json > /dev/null
python3 tools/valu_mix.py profiles/${R}_k_accumulate29_valu_mix.json > /dev/null
[ -f gpurun_out/${R}_ntt_pmc.json ] && cp gpurun_out/${R}_ntt_pmc.json profiles/${R}_ntt_pmc.json && cp gpurun_out/${R}_ntt_kernel_stats.csv profiles/${R}_ntt_kernel_stats.csv
python3 tools/valu_mix.py profiles/${R}_k_ntt_pass_valu_mix.json --kernel ntt > /dev/null
[ -f gpurun_out/${R}_bench_n1.json ] && tail -1 gpurun_out/${R}_bench_n1.json > profiles/${R}_bench_line.json
[ -f gpurun_out/${R}_gates_pmc.json ] && cp gpurun_out/${R}_gates_pmc.json gpurun_out/${R}_gates_kernel_stats.csv profiles/
for f in prover_native gate_kernels lookup_prover kmin_sweep prover_kernel_stats.csv prover_timeline; do
  for g in gpurun_out/${R}_${f}.txt gpurun_out/${R}_${f}; do [ -f "$g" ] && cp "$g" profiles/; done
done
echo installed profiles/${R}_*
