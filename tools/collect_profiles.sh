#!/bin/bash
# Collects the round's committed evidence on the GPU box (run from the repo root through gpurun): rocprofv3 kernel stats + PMC passes for the MSM and the
# NTT workloads (tools/profile_msm.py: every PMC pass its own run, never combined with a trace), the pipelined kernel trace + timeline, the static
# opcode mixes, the A/B of the two table sets, the PCIe-inclusive rates, and finally bench.py's line (which then finds profiles of ITS OWN build).
# Usage: tools/collect_profiles.sh r06
R=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python tools/profile_msm.py gpurun_out/prof/${R}_msm20 > gpurun_out/prof/${R}_msm20_summary.txt 2>&1
python tools/profile_msm.py gpurun_out/prof/${R}_ntt --workload ntt > gpurun_out/prof/${R}_ntt_summary.txt 2>&1
python tools/valu_mix.py gpurun_out/prof/${R}_k_acc_wide29_valu_mix.json > /dev/null 2>&1
python tools/valu_mix.py gpurun_out/prof/${R}_k_ntt_pass_valu_mix.json --kernel ntt > /dev/null 2>&1
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/pipe_$R -o t -- python $OLDPWD/tools/msm_loop.py wide pipe 40 20 2 > /dev/null 2>&1)
DB=$(find /tmp/pipe_$R -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/prof/${R}_msm20_pipelined_kernel_stats.csv > /dev/null
python tools/timeline.py $DB > gpurun_out/prof/${R}_msm20_pipelined_timeline.txt
python tools/wide_ab.py 20 20 5 > gpurun_out/prof/${R}_wide_ab.txt 2>&1
python tools/pcie_inclusive.py > gpurun_out/prof/${R}_pcie_inclusive.txt 2>&1
python tools/bench_ntt.py > gpurun_out/prof/${R}_bench_ntt.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/proof_$R -o t -- python $OLDPWD/tools/prover_time.py 16 --native > /dev/null 2>&1)
PDB=$(find /tmp/proof_$R -name "*.db" | head -1)
python tools/proof_timeline.py $PDB > gpurun_out/prof/${R}_proof_timeline.txt 2>&1
python tools/round_timeline.py $PDB > gpurun_out/prof/${R}_round_timeline.txt 2>&1
python tools/proof_busy.py $PDB gpurun_out/prof/${R}_proof_busy.json > /dev/null 2>&1
KH_IPA_TIMING=1 python tools/prover_time.py 16 --native > gpurun_out/prof/${R}_prover_time_ipa_timing.txt 2>&1
# bench.py last: copy the fresh PMC / mix files where it looks for them
cp gpurun_out/prof/${R}_proof_busy.json gpurun_out/prof/${R}_msm20_pmc.json gpurun_out/prof/${R}_ntt_pmc.json gpurun_out/prof/${R}_k_acc_wide29_valu_mix.json gpurun_out/prof/${R}_k_ntt_pass_valu_mix.json profiles/ 2>/dev/null
python bench.py > gpurun_out/prof/${R}_bench_stdout.txt 2> gpurun_out/prof/${R}_bench_stderr.txt
grep '^{' gpurun_out/prof/${R}_bench_stdout.txt | tail -1 > gpurun_out/prof/${R}_bench_line.json
rm -rf gpurun_out/prof/*_passes
ls -la gpurun_out/prof | head -40
