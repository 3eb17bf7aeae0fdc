#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/dbg_goff.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c3_dbg.log
timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -s \
    -k "bench_shape or split6 or deterministic or conv_split6" > gpurun_out/c3_pytest_a.log 2>&1
echo "pytest A rc $?"; grep -E "passed|failed|^FAILED" gpurun_out/c3_pytest_a.log | tail -12
