#!/bin/bash
# round 6: the residue classes of a strided data gradient in ONE launch (LSNET_BWD_MERGE=0: one launch per class, as before);
# partner ab_v10.so = the commit before (the kernel without per-level geometry)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resblock_gpu.py -q -m gpu -x -k "conv or resblock or bottleneck" 2>&1 | tail -3
out=gpurun_out/r6_bwd_merge.txt
: > $out
for rep in 1 2; do
  echo "== merged (rep $rep)" >> $out
  timeout 100 tools/ubench/conv_step 10 2>&1 | grep -E " s2 |per step" >> $out
  echo "== LSNET_BWD_MERGE=0 (rep $rep)" >> $out
  LSNET_BWD_MERGE=0 timeout 100 tools/ubench/conv_step 10 2>&1 | grep -E " s2 |per step" >> $out
  echo "== previous commit (rep $rep)" >> $out
  LSNET_SO=lsnet_amd/csrc/ab_v10.so timeout 100 tools/ubench/conv_step 10 2>&1 | grep -E " s2 |per step" >> $out
done
cat $out | cut -c1-120
bash tools/r6_calls/c26_ab.sh ab_v10 r6_bwd_merge_step | grep -E "^old|^new"
