#!/bin/bash
# round 6: stream-K pieces in the deformable forward (A/B through debug bit 19) and the stream-order matrix with submissions
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 2; do
  for dbg in 0 524288; do
    echo "== DCN_STEP_DBG=$dbg (524288 = whole tiles only, round 5's distribution)"
    DCN_STEP_DBG=$dbg timeout 120 tools/ubench/dcn_step both 10 2>&1 | grep -v "debug bit 28" | grep -E "default kernels|dcn_fwd|against the host|backward twice"
  done
done > gpurun_out/r6_dcn_sk.txt 2>&1
cat gpurun_out/r6_dcn_sk.txt
timeout 1500 python tools/rccl_streams.py > gpurun_out/r6_rccl_streams2.txt 2>&1; echo "streams rc $?"
cat gpurun_out/r6_rccl_streams2.txt
