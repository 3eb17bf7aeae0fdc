#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_golden_gpu.py tests/test_zz_cpv_gpu.py tests/test_variants_gpu.py -q -m gpu -s > gpurun_out/r6_c10_golden.log 2>&1; echo "golden rc $?"
grep -E "passed|failed|Error" gpurun_out/r6_c10_golden.log | tail -5
grep -E "curve|FAILED" gpurun_out/r6_c10_golden.log | cut -c1-330 | tail -50
