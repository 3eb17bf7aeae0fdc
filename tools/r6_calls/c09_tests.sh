#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -s -k "exact_mode or deterministic or split6 or math_fp32 or fp32" > gpurun_out/r6_c09_ops.log 2>&1; echo "ops rc $?"
grep -E "passed|failed" gpurun_out/r6_c09_ops.log | tail -3
timeout 2400 python -m pytest tests/test_golden_gpu.py tests/test_zz_cpv_gpu.py tests/test_variants_gpu.py -q -m gpu -x -s > gpurun_out/r6_c09_golden.log 2>&1; echo "golden rc $?"
grep -E "passed|failed|Error" gpurun_out/r6_c09_golden.log | tail -5
grep -E "curve" gpurun_out/r6_c09_golden.log | cut -c1-330 | tail -40
