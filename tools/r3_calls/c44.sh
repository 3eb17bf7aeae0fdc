#!/bin/bash
# the DENSE weight-gradient kernel in the 3-product mode, then the dense-convolution parity tests with the new default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LSNET_MATH=bf16x3 LSNET_CONV_WGRAD_MM=1 timeout 30 tools/ubench/wgrad_ab > gpurun_out/c44_wgrad_ab_x3.log 2>&1
echo "rc $?"; tail -8 gpurun_out/c44_wgrad_ab_x3.log | cut -c1-140
timeout 48 python -m pytest tests/test_ops_gpu.py -x -q -p no:cacheprovider -k "split6 or conv2d_matches_torch or multi_level_equals" 2>&1 | tail -4
