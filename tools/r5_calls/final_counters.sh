#!/bin/bash
# round 5, final measurement call A: HBM counters of the deformable launches (FETCH / WRITE, separate passes), SQ counters, HBM
# counters of the stream kernel classes over the whole step, kernel trace of the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/pmc_step_shapes.sh r5 > gpurun_out/r5_pmc_step.log 2>&1
bash tools/pmc_sq_step_shapes.sh r5 > gpurun_out/r5_pmc_sq_step.log 2>&1
bash tools/pmc_stream_step.sh r5 > gpurun_out/r5_pmc_stream_step.log 2>&1
bash tools/profile_bench.sh r5b 3 --no-extra > gpurun_out/r5b_profile.log 2>&1
tail -12 gpurun_out/r5_pmc_hbm.txt; head -8 gpurun_out/r5b_kernel_stats.txt | cut -c1-160
