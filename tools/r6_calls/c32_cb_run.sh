#!/bin/bash
# round 6: runs of three column blocks per workgroup in the deformable backward GEMM (debug bit 16 = 65536: one block, as before)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r6_cb_run.txt
: > $out
for rep in 1 2; do
  for dbg in 65536 0; do
    echo "== DCN_STEP_DBG=$dbg" >> $out
    DCN_STEP_DBG=$dbg timeout 100 tools/ubench/dcn_step both 10 2>&1 | grep -v "debug bit 28" | grep -E "default kernels|against the host|backward twice|dcn_bwd_data" >> $out
  done
done
grep -E "^==|dcn_bwd_data" $out | awk '/^==/{name=$0; n=0; next} {n++; if(n==1) printf "%-28s tower %s", name, $4; if(n==3) printf "   pyramid %s\n", $4}'
grep -E "grad_input|differ" $out | head -4 | cut -c100-260
