#!/bin/bash
# ordered reduce in round 2's weight-gradient kernel (strided 3x3 dense convs, deformable shapes outside dcn_mm_kernels.h)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "conv2d_matches or (test_dcn_forward_backward and (default or first_gemms or x3_gather)) or multi_level" 2>&1 | tail -2
timeout 300 python tools/repro_step.py 2>&1 | grep -E "^loss|parameter gradients differ|^  [a-z]" | head -12
timeout 300 python tools/repro_step.py 384 480 2>&1 | grep -E "^loss|parameter gradients differ|^  [a-z]" | head -12
