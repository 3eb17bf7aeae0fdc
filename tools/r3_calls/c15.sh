#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v "amdgpu\|Warn\|warn\|got = " > gpurun_out/c15_gpu_tests.log
grep -E "same device|passed|failed|FAILED|Error" gpurun_out/c15_gpu_tests.log | tail -30
