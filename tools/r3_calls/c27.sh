#!/bin/bash
# full -m gpu suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -v "amdgpu\|UserWarning\|warnings.warn\|got = " > gpurun_out/c27_gpu_tests.log
grep -E "passed|failed|FAILED|Error" gpurun_out/c27_gpu_tests.log | tail -15
