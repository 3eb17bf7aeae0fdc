#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -q --tb=short -p no:cacheprovider -k "bench_shape or deterministic or test_dcn_forward_backward or training_curve" > gpurun_out/c18_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c18_pytest.log | tail -5
