#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids > gpurun_out/c09_convs.log
tail -32 gpurun_out/c09_convs.log | cut -c1-160
echo "== old wgrad kernel"
LSNET_WGRAD_OLD=1 timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids | tail -32 | cut -c1-41,72-90
