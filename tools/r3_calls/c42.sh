#!/bin/bash
# A/B of the dense weight gradient through dcn_wgrad_mm_kernel<NP, DENSE> (no torch: the whole call is a few seconds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 60 tools/ubench/wgrad_ab > gpurun_out/c42_wgrad_ab.log 2>&1
echo "rc $?"
cat gpurun_out/c42_wgrad_ab.log
