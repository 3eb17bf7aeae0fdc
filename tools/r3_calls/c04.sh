#!/bin/bash
# workspace split-K: operator tests, probe, per-layer table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "conv" 2>&1 | tail -5 > gpurun_out/c04_tests.log
tail -3 gpurun_out/c04_tests.log
SH="l3_3x3:256:256:3:1:50:84 l3_1x1a:256:1024:1:1:50:84 l3_1x1b:1024:256:1:1:50:84 l4_3x3:512:512:3:1:25:42 l4_1x1a:512:2048:1:1:25:42 l4_1x1b:2048:512:1:1:25:42 l4_1x1c:1024:512:1:1:50:84 P6:2048:256:3:2:25:42 P5lat:2048:256:1:1:25:42 off_P4:256:27:3:1:50:84 off_P3:256:27:3:1:100:168 l2_3x3:128:128:3:1:100:168 P7:256:256:3:2:13:21"
for KS in 0 1 2 4 8; do
  echo "== LSNET_CONV_KSPLIT=$KS"
  LSNET_CONV_KSPLIT=$KS timeout 100 python tools/conv_probe.py $SH 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c04_ksplit.log 2>&1
cat gpurun_out/c04_ksplit.log
timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids > gpurun_out/c04_convs.log
tail -32 gpurun_out/c04_convs.log | cut -c1-150
