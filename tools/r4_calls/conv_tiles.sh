#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for t in 0 1 2 5 7 8 9 10; do
    echo "== LSNET_CONV_TILE=$t (0: the library's choice)" >> gpurun_out/r4_conv_tiles.log
    LSNET_CONV_TILE=$t timeout 40 tools/ubench/conv_step 10 >> gpurun_out/r4_conv_tiles.log 2>&1
done
for k in 1 2 4 8; do
    echo "== LSNET_CONV_KSPLIT=$k" >> gpurun_out/r4_conv_tiles.log
    LSNET_CONV_KSPLIT=$k timeout 40 tools/ubench/conv_step 10 >> gpurun_out/r4_conv_tiles.log 2>&1
done
grep "==\|per step" gpurun_out/r4_conv_tiles.log
# the deformable replay with the backward GEMM (a 1x1 convolution with N = 2304) on the fat tiles
for t in 0 7 8 9; do
    echo "== dcn_step, LSNET_CONV_TILE=$t" >> gpurun_out/r4_conv_tiles.log
    LSNET_CONV_TILE=$t timeout 60 tools/ubench/dcn_step tower 5 2>&1 | grep -v "debug bit 28" | head -12 >> gpurun_out/r4_conv_tiles.log
done
grep -A8 "dcn_step" gpurun_out/r4_conv_tiles.log | grep "==\|backward\|dcn_bwd_data\|against"
