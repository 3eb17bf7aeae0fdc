#!/bin/bash
# round 5: conv_mm_kernel main loop with the barrier in the middle of the iteration (product) vs round 4's loop (A/B library
# built with -DLSNET_OLD_LOOP), same stream-K distribution in both: conv_step + tile_sweep + dcn_step (its backward GEMM), one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
    echo "== $so"
    LSNET_SO=$so timeout 90 tools/ubench/conv_step 10; echo "rc $?"
    LSNET_SO=$so timeout 90 tools/ubench/dcn_step tower 5 | grep -E "default kernels|dcn_bwd_data|against" | head -3
  done
done > gpurun_out/r5_c07_loop.log 2>&1
grep -E "^==|per step|dcn_bwd|default kernels|rc " gpurun_out/r5_c07_loop.log
