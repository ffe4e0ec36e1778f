#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
bash scripts/trace_run.sh r05e_c5_sync --no-drop-in --sync-steps --workload c5 --views 200 > /dev/null 2>&1
bash scripts/trace_run.sh r05e_sync --no-drop-in --sync-steps > /dev/null 2>&1
bash scripts/pmc_profile.sh r05e_pmc > gpurun_out/r05e_pmc.log 2>&1
cat gpurun_out/r05e_c5_sync_kernel_stats.txt | head -24
cat gpurun_out/r05e_sync_kernel_stats.txt | head -24
cat gpurun_out/r05e_pmc.txt | head -24
