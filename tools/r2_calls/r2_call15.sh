#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_zz_fused_ciou_gpu.py tests/test_golden_gpu.py tests/test_variants_gpu.py -q -s --tb=short -p no:cacheprovider -k "fused or head_forward or one_training_step" > gpurun_out/c15_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error|worst" gpurun_out/c15_pytest.log | tail -16
