#!/bin/bash
# round 6: TRANS epilogue of conv_mm_kernel (the backward GEMM of the deformable family) through a buffer descriptor
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/r6_calls/c18_anchor_order.sh r6_trans_epi ab_v5 > /dev/null
grep -E "^==|dcn_bwd_data|against|twice" gpurun_out/r6_trans_epi.txt | grep -v "old" | cut -c1-210
