#!/bin/bash
# launch census of the step as of the 64x256 tile / per-anchor gather
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/profile_bench.sh c23 3 --no-extra
head -75 gpurun_out/c23_kernel_stats.txt | cut -c1-150
