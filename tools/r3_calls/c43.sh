#!/bin/bash
# the library's own selection between the two dense weight-gradient kernels (LSNET_CONV_WGRAD_MM unset = 2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LSNET_CONV_WGRAD_MM=2 timeout 60 tools/ubench/wgrad_ab > gpurun_out/c43_wgrad_rule.log 2>&1
echo "rc $?"
cat gpurun_out/c43_wgrad_rule.log
