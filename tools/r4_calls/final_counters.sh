#!/bin/bash
# round 4, final measurement call A: HBM counters (FETCH / WRITE, separate passes), SQ counters, kernel trace of the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/pmc_step_shapes.sh r4 > gpurun_out/r4_pmc_step.log 2>&1
bash tools/pmc_sq_step_shapes.sh r4 > gpurun_out/r4_pmc_sq_step.log 2>&1
bash tools/profile_bench.sh r4d 3 --no-extra > gpurun_out/r4d_profile.log 2>&1
tail -12 gpurun_out/r4_pmc_hbm.txt; head -8 gpurun_out/r4d_kernel_stats.txt | cut -c1-160
