#!/bin/bash
# The three torch-free harnesses in one gpurun call (a few seconds of GPU time):
#   gpurun --timeout 120 -- 'bash tools/ubench/run_all.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for t in wgrad_ab dcn_step conv_step norm_step; do
    [ -x tools/ubench/$t ] || hipcc --offload-arch=gfx950 -O2 tools/ubench/$t.hip -o tools/ubench/$t -ldl
done
timeout 60 tools/ubench/dcn_step both 5 > gpurun_out/ubench_dcn_step.log 2>&1; echo "dcn_step rc $?"
timeout 60 tools/ubench/conv_step 10 > gpurun_out/ubench_conv_step.log 2>&1; echo "conv_step rc $?"
timeout 60 tools/ubench/norm_step 10 > gpurun_out/ubench_norm_step.log 2>&1; echo "norm_step rc $?"
LSNET_CONV_WGRAD_MM=1 timeout 60 tools/ubench/wgrad_ab > gpurun_out/ubench_wgrad_ab.log 2>&1; echo "wgrad_ab rc $?"
tail -n 40 gpurun_out/ubench_dcn_step.log
tail -n 3 gpurun_out/ubench_conv_step.log gpurun_out/ubench_wgrad_ab.log
cat gpurun_out/ubench_norm_step.log
