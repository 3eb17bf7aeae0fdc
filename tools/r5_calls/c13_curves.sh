#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden_gpu.py -q -m gpu -s -k "training_curve" > gpurun_out/r5_c13_curves.log 2>&1; echo "rc $?"
grep -E "curve loss|curve loss_|worst|passed|failed|Error" gpurun_out/r5_c13_curves.log | cut -c1-400
