#!/bin/bash
# round 5: the mid-iteration barrier in dcn_fwd_mm_kernel / dcn_wgrad_mm_kernel (product) vs the round-4 loops (A/B library), dcn_step + wgrad_ab (dense layers on the deformable kernel), one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
    echo "== $so"
    LSNET_SO=$so timeout 120 tools/ubench/dcn_step both 5 | grep -E "default kernels|dcn_fwd |dcn_wgrad |dcn_bwd_data |against the host|backward twice" | grep -v "debug bit" | head -12
    timeout 90 tools/ubench/wgrad_ab $so rule | tail -1
  done
done > gpurun_out/r5_c11_dcnloop.log 2>&1
cat gpurun_out/r5_c11_dcnloop.log
