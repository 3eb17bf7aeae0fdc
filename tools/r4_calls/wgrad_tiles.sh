#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for t in 0 11 12 21 22; do
    echo "== LSNET_WGRAD_TILE=$t (0: the library's choice)" >> gpurun_out/r4_wgrad_tiles.log
    LSNET_CONV_WGRAD_MM=0 LSNET_WGRAD_TILE=$t timeout 40 tools/ubench/wgrad_ab 2>&1 | cut -c1-70 >> gpurun_out/r4_wgrad_tiles.log
done
grep "==\|per step" gpurun_out/r4_wgrad_tiles.log
