#!/bin/bash
# round 4, call 2: the adopted defaults (fine interleave in all four families, new tile rule) against the A/B build with
# the fine form for the narrow dense tiles as well; the full harness set on the new defaults; kernel trace of dcn_step (csv)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r4_c02.log
: > $L
for so in lsnet_amd/csrc/liblsnet_hip.so lsnet_amd/csrc/liblsnet_hip_ab.so; do
  for rep in 1 2; do
    echo "== conv_step $so rep $rep" >> $L
    LSNET_SO=$PWD/$so timeout 60 tools/ubench/conv_step 10 >> $L 2>&1
  done
done
echo "== dcn_step" >> $L
timeout 60 tools/ubench/dcn_step both 5 2>&1 | grep -v "^    default vs old" >> $L
echo "== wgrad_ab" >> $L
LSNET_CONV_WGRAD_MM=1 timeout 60 tools/ubench/wgrad_ab >> $L 2>&1
grep "==\|per step\|launches\|against" $L
export TMPDIR=/tmp
R="$PWD"
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4_dcn_trace2" -o dcn_step -- "$R/tools/ubench/dcn_step" both 3 > "$R/gpurun_out/r4_dcn_trace2.log" 2>&1)
find gpurun_out/r4_dcn_trace2 -name "*stats*" | head
