#!/bin/bash
# round 6: more, shorter workgroups in the grad_output fragment pre-pass of the fragment-order weight gradients
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r6_spb.txt
: > $out
for rep in 1 2; do
  for v in liblsnet_hip ab_spb1024 ab_spb2048; do
    echo "== $v (rep $rep)" >> $out
    timeout 100 tools/ubench/wgrad_ab lsnet_amd/csrc/$v.so rule 2>&1 | tail -1 >> $out
    LSNET_SO=lsnet_amd/csrc/$v.so timeout 100 tools/ubench/dcn_step both 10 2>&1 | grep -v "debug bit 28" | grep -E "dcn_wgrad" | sed -n '1p;3p' >> $out
  done
done
cat $out
