#!/bin/bash
# round 5: band pipeline of the deformable backward (column-gradient GEMM band after band, per-anchor sums of band i beside the
# GEMM of band i + 1) against the one-launch form (debug bit 20 = 1048576), tools/ubench/dcn_step, alternating on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  echo "== bands"; timeout 120 tools/ubench/dcn_step both 5; echo "rc $?"
  echo "== one launch (debug bit 20)"; DCN_STEP_DBG=1048576 timeout 120 tools/ubench/dcn_step both 5; echo "rc $?"
done > gpurun_out/r5_c05_bands.log 2>&1
grep -v "^$" gpurun_out/r5_c05_bands.log | tail -n 80
