#!/bin/bash
# full GPU test suite (the driver's round-end command) with per-test durations and the printed deviations kept
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider --durations=25 > gpurun_out/r2_gpu_tests.log 2>&1
echo "pytest rc $?"
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r2_gpu_tests.log | tail -15
grep -E "^curve |worst|deviation" gpurun_out/r2_gpu_tests.log | head -60
