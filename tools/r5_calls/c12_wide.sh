#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
    echo "== $so"
    LSNET_SO=$so timeout 90 tools/ubench/conv_step 10; echo "rc $?"
  done
done > gpurun_out/r5_c12_wide.log 2>&1
grep -E "^==|per step" gpurun_out/r5_c12_wide.log
