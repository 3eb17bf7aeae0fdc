#!/bin/bash
# round 5, call 1: stream-K work distribution of conv_mm_kernel (product) against rounds 3/4's z-split + tail split (A/B
# library), tools/ubench/conv_step, alternating on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
    echo "== $so"
    LSNET_SO=$so timeout 90 tools/ubench/conv_step 10; echo "rc $?"
  done
done > gpurun_out/r5_c01_sk.log 2>&1
tail -n 75 gpurun_out/r5_c01_sk.log
