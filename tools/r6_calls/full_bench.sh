#!/bin/bash
# round 6: the default bench line (extra legs + CPU baseline), as the driver runs it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/r6_bench.log 2>&1
tail -c 3000 gpurun_out/r6_bench.log
