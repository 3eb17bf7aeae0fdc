#!/bin/bash
# round 6: buffer-descriptor stores of the partial tiles in conv_wgrad_kernel; partner ab_v7.so = the commit before
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resblock_gpu.py tests/test_zz_grad_sink_gpu.py -q -m gpu -x -k "conv or resblock or bottleneck or wgrad or weight_grad or deferred" 2>&1 | tail -3
bash tools/r6_calls/c26_ab.sh ab_v7 r6_wgrad_epi | grep -E "^==|wgrad_ab|^old|^new"
