#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for t in 0 1 2 5; do
    echo "== LSNET_CONV_TILE=$t (0: the library's choice)" >> gpurun_out/r4_conv_tiles.log
    LSNET_CONV_TILE=$t timeout 40 tools/ubench/conv_step 10 >> gpurun_out/r4_conv_tiles.log 2>&1
done
for k in 1 2 4 8; do
    echo "== LSNET_CONV_KSPLIT=$k" >> gpurun_out/r4_conv_tiles.log
    LSNET_CONV_KSPLIT=$k timeout 40 tools/ubench/conv_step 10 >> gpurun_out/r4_conv_tiles.log 2>&1
done
grep "==\|per step" gpurun_out/r4_conv_tiles.log
