#!/bin/bash
# round 6: config 4 (X-101-64x4d-DCN segm) with a partner library (LSNET_HIP_SO) and the product one, alternating; grouped tests first
# usage: c38_grouped.sh <partner .so name> <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
old=${1:-ab_v11}; tag=${2:-r6_grouped}
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "grouped or x101 or dcn_forward_backward" 2>&1 | tail -2
out=gpurun_out/$tag.txt
: > $out
for rep in 1 2; do
  for v in $old liblsnet_hip; do
    echo "== $v (rep $rep)" >> $out
    LSNET_HIP_SO=$PWD/lsnet_amd/csrc/$v.so timeout 600 python tools/config_steps.py segm x101-dcn 3 2>&1 | grep -E "ms/step" | head -8 >> $out
  done
done
cat $out
