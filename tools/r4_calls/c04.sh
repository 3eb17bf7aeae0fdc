#!/bin/bash
# round 4, call 4: folded-norm weight gradient vs the plain one (reduce cost), whole-line stores in the deformable
# backward GEMM (default build) vs lane-per-pixel stores (A/B build), relu gate pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r4_c04.log
: > $L
echo "== wgrad_ab bn" >> $L
timeout 90 tools/ubench/wgrad_ab lsnet_amd/csrc/liblsnet_hip.so bn >> $L 2>&1
for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
  for rep in 1 2; do
    echo "== dcn_step $so rep $rep" >> $L
    LSNET_SO=$PWD/$so timeout 60 tools/ubench/dcn_step both 5 2>&1 | grep -v "debug bit 28\|default vs old" | grep "tower\|pyramid\|dcn_bwd_data\|against\|twice" >> $L
  done
done
echo "== norm_step" >> $L
timeout 60 tools/ubench/norm_step 10 >> $L 2>&1
cat $L
