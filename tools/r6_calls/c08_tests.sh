#!/bin/bash
# round 6: after the kernel clean-up (windowed / pipelined kernels gone, exact mode on the gather path): operator, golden, CPV, variant tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_zz_cpv_gpu.py tests/test_variants_gpu.py -q -m gpu -x -s > gpurun_out/r6_c08_tests.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error|error" gpurun_out/r6_c08_tests.log | tail -8
grep -E "curve|benchmark-model" gpurun_out/r6_c08_tests.log | tail -40
