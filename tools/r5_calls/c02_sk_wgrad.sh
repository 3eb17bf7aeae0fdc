#!/bin/bash
# round 5, call 4: stream-K policy v4 (conv_step) + weight-gradient split count rounded down (wgrad_ab rule / bn), product vs A/B library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
    echo "== $so"
    LSNET_SO=$so timeout 90 tools/ubench/conv_step 10; echo "rc $?"
  done
done > gpurun_out/r5_c02_sk.log 2>&1
for i in 1 2; do
  for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
    echo "== $so rule"; timeout 90 tools/ubench/wgrad_ab $so rule; echo "rc $?"
    echo "== $so bn"; timeout 90 tools/ubench/wgrad_ab $so bn; echo "rc $?"
  done
done > gpurun_out/r5_c02_wgrad.log 2>&1
grep -E "^==|per step|rc " gpurun_out/r5_c02_sk.log
tail -n 60 gpurun_out/r5_c02_wgrad.log
