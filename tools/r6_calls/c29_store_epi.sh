#!/bin/bash
# round 6: branch-free buffer stores in both epilogues of conv_mm_kernel; partner ab_v5.so = the commit before
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resblock_gpu.py -q -m gpu -x -k "conv or resblock or bottleneck or dcn_forward_backward or tower_launch or pyramid_launch" 2>&1 | tail -3
bash tools/r6_calls/c26_ab.sh ab_v5 r6_store_epi | grep -E "^==|per step|^old|^new|dcn_bwd_data"
