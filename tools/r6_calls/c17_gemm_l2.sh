#!/bin/bash
# round 6: backward GEMM of the deformable family -- nontemporal column-gradient stores (nt), supertile order of
# (pixel tile, column block) (cg: 16 pixel tiles x 3 column blocks), both (ntcg) against the product library.
# Built by: python tools/build_variants.py nt:-DLSNET_AB_NT cg:-DLSNET_AB_COLGRP=3,-DLSNET_AB_PT=16 ntcg:-DLSNET_AB_NT,-DLSNET_AB_COLGRP=3,-DLSNET_AB_PT=16
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r6_gemm_l2.txt
: > $out
for rep in 1 2; do
  for v in liblsnet_hip ab_nt ab_cg ab_ntcg; do
    echo "== $v (rep $rep)" >> $out
    LSNET_SO=lsnet_amd/csrc/$v.so timeout 120 tools/ubench/dcn_step both 10 2>&1 | grep -v "debug bit 28" | grep -E "default kernels|against the host|backward twice" >> $out
  done
done
for v in liblsnet_hip ab_nt ab_cg ab_ntcg; do
  raw=/tmp/prof_$v; rm -rf $raw
  LSNET_SO=lsnet_amd/csrc/$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $raw -o t -- tools/ubench/dcn_step both 10 > /dev/null 2>&1
  echo "== $v kernel stats" >> $out
  f=$(find $raw -name "*kernel_stats.csv" | head -1)
  grep -E "conv_mm_kernel|anchor|dcn_wgrad_mm|dcn_fwd_mm" $f | cut -c1-200 >> $out
done
cat $out
