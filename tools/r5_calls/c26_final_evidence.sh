#!/bin/bash
# round 5, final tree: kernel trace of the timed steps, ATen launches by call site, then the default bench line as the driver runs it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/profile_bench.sh r5 3 --no-extra > gpurun_out/r5_profile.log 2>&1
head -3 gpurun_out/r5_kernel_stats.txt
timeout 400 python tools/torch_prof_sites.py > gpurun_out/r5_aten_sites.txt 2>&1
grep "ATen ops" gpurun_out/r5_aten_sites.txt
timeout 1200 python bench.py > gpurun_out/r5_bench.log 2>&1
tail -c 2500 gpurun_out/r5_bench.log
