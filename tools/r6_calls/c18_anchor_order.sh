#!/bin/bash
# round 6: per-anchor sums / combine walk their anchors (pixels) in REVERSE order -- the column gradients the GEMM wrote
# last are still in the Infinity Cache (rev) --, nontemporal loads of the column-gradient rows (ntl), both (revntl);
# the product library (nontemporal GEMM stores adopted) as the partner.
# Built by: python tools/build_variants.py rev:-DLSNET_AB_REV ntl:-DLSNET_AB_NTL revntl:-DLSNET_AB_REV,-DLSNET_AB_NTL
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/${1:-r6_anchor_order}.txt
shift
: > $out
for rep in 1 2; do
  for v in liblsnet_hip "$@"; do
    echo "== $v (rep $rep)" >> $out
    LSNET_SO=lsnet_amd/csrc/$v.so timeout 100 tools/ubench/dcn_step both 10 2>&1 | grep -v "debug bit 28" | grep -E "default kernels|against the host|backward twice|dcn_bwd_data|dcn_wgrad" | grep -v "^    dcn_.*old" >> $out
  done
done
cat $out | cut -c1-200
