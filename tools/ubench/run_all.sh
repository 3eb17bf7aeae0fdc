#!/bin/bash
# The torch-free harnesses in one gpurun call (a few seconds of GPU time).  Build them first, in the container:
#   bash tools/ubench/build.sh && gpurun --timeout 120 -- 'bash tools/ubench/run_all.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 60 tools/ubench/dcn_step both 5 > gpurun_out/ubench_dcn_step.log 2>&1; echo "dcn_step rc $?"
timeout 60 tools/ubench/conv_step 10 > gpurun_out/ubench_conv_step.log 2>&1; echo "conv_step rc $?"
timeout 60 tools/ubench/norm_step 10 > gpurun_out/ubench_norm_step.log 2>&1; echo "norm_step rc $?"
timeout 60 tools/ubench/wgrad_ab lsnet_amd/csrc/liblsnet_hip.so rule > gpurun_out/ubench_wgrad_ab.log 2>&1; echo "wgrad_ab rc $?"
timeout 60 tools/ubench/wgrad_ab lsnet_amd/csrc/liblsnet_hip.so bn > gpurun_out/ubench_wgrad_bn.log 2>&1; echo "wgrad_ab bn rc $?"
timeout 60 tools/ubench/tile_sweep 256 > gpurun_out/ubench_tile_sweep.log 2>&1; echo "tile_sweep rc $?"
tail -n 40 gpurun_out/ubench_dcn_step.log
tail -n 3 gpurun_out/ubench_conv_step.log gpurun_out/ubench_wgrad_ab.log gpurun_out/ubench_wgrad_bn.log
cat gpurun_out/ubench_norm_step.log
