#!/bin/bash
# round 6, final measurement call A: HBM counters of the deformable launches (FETCH / WRITE, separate passes), SQ counters, HBM
# counters of the stream kernel classes over the whole step, kernel trace of the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/pmc_step_shapes.sh r6 > gpurun_out/r6_pmc_step.log 2>&1
bash tools/pmc_sq_step_shapes.sh r6 > gpurun_out/r6_pmc_sq_step.log 2>&1
bash tools/pmc_stream_step.sh r6 > gpurun_out/r6_pmc_stream_step.log 2>&1
bash tools/profile_bench.sh r6 3 --no-extra > gpurun_out/r6_profile.log 2>&1
tail -12 gpurun_out/r6_pmc_hbm.txt; head -8 gpurun_out/r6_kernel_stats.txt | cut -c1-160
