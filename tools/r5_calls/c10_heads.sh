#!/bin/bash
# head fixtures with the shifted norm biases (tests/golden_util.py HEAD_NORM_BIAS_SHIFT) on the device: measured deviations (-s)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_zz_cpv_gpu.py -q -m gpu -s -k "head_forward or head_at_256 or cpv or decode_is" > gpurun_out/r5_c10_heads.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|worst deviation|gradient checks|Error" gpurun_out/r5_c10_heads.log | cut -c1-330
