#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh r03'): every measurement profiles/ quotes, on one build, into gpurun_out/.
# Afterwards, here: bash tools/install_profiles.sh r03
R=${1:-r03}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tools/microbench > gpurun_out/${R}_microbench.txt 2>&1
python tools/profile_msm.py gpurun_out/${R}_msm20 > gpurun_out/${R}_profile_msm.log 2>&1
python tools/profile_msm.py gpurun_out/${R}_ntt --workload ntt > gpurun_out/${R}_profile_ntt.log 2>&1
python tools/profile_msm.py gpurun_out/${R}_gates --workload gates > gpurun_out/${R}_profile_gates.log 2>&1
python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench_n1.err
( cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace -d /tmp/pp_${R} -o pp -- python $GRAFT_REPO_ROOT/tools/prover_time.py 16 > $GRAFT_REPO_ROOT/gpurun_out/${R}_prover_time.txt 2>&1 )
python tools/rocpd_stats.py /tmp/pp_${R}/pp_results.db gpurun_out/${R}_prover_kernel_stats.csv > /dev/null 2>&1
python tools/prover_timeline.py /tmp/pp_${R}/pp_results.db --opening > gpurun_out/${R}_prover_timeline.txt 2>&1
# the measurements DESIGN.md quotes beside the standard set: native prover latency / threads, gate kernels against the token machine, lookup proofs, task-length sweep
python tools/prover_time.py 16 --native > gpurun_out/${R}_prover_native.txt 2>&1
python tools/prover_concurrent.py 16 1 2 4 6 --native >> gpurun_out/${R}_prover_native.txt 2>&1
python tools/gate_expr_time.py > gpurun_out/${R}_gate_kernels.txt 2>&1
python tools/lookup_prover_time.py 16 > gpurun_out/${R}_lookup_prover.txt 2>&1
bash tools/kmin_sweep.sh > gpurun_out/${R}_kmin_sweep.txt 2>&1
tail -3 gpurun_out/${R}_prover_time.txt; tail -c 600 gpurun_out/${R}_bench_n1.json; ls -la gpurun_out | tail -20
