#!/bin/bash
# split-K sweep on the small-P / deep-K shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
SH="l3_3x3:256:256:3:1:50:84 l3_1x1a:256:1024:1:1:50:84 l3_1x1b:1024:256:1:1:50:84 l4_3x3:512:512:3:1:25:42 l4_1x1a:512:2048:1:1:25:42 l4_1x1b:2048:512:1:1:25:42 l4_1x1c:1024:512:1:1:50:84 P6:2048:256:3:2:25:42 P5lat:2048:256:1:1:25:42 off_P4:256:27:3:1:50:84 off_P3:256:27:3:1:100:168 l2_3x3:128:128:3:1:100:168"
for KS in 0 1 2 3 4 6 8; do
  echo "== LSNET_CONV_KSPLIT=$KS"
  LSNET_CONV_KSPLIT=$KS timeout 100 python tools/conv_probe.py $SH 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c03_ksplit.log 2>&1
for TL in 1 2; do
  echo "== LSNET_CONV_TILE=$TL"
  LSNET_CONV_TILE=$TL timeout 100 python tools/conv_probe.py l3_3x3:256:256:3:1:50:84 l2_3x3:128:128:3:1:100:168 l2_1x1:128:512:1:1:100:168 P3:256:256:3:1:100:168 all5:256:256:3:1:140:160 2>&1 | grep -v amdgpu.ids
done >> gpurun_out/c03_ksplit.log 2>&1
cat gpurun_out/c03_ksplit.log
