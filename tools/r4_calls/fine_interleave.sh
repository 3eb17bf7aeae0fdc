#!/bin/bash
# The fine MFMA / staging interleave (slice commits without their branch + sched_group_barrier) in the four kernel
# families, each against its default form on the launch shapes of the step.  All torch-free.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r4_fine.log
: > $L
for v in 0 1; do
    echo "== dense weight gradient, LSNET_WGRAD_FINE=$v (patch kernel only)" >> $L
    LSNET_CONV_WGRAD_MM=0 LSNET_WGRAD_FINE=$v timeout 40 tools/ubench/wgrad_ab 2>&1 | cut -c1-66 >> $L
    echo "== deformable launches, LSNET_DCN_FWD_FINE=$v LSNET_DCN_WGRAD_FINE=$v" >> $L
    LSNET_DCN_FWD_FINE=$v LSNET_DCN_WGRAD_FINE=$v timeout 60 tools/ubench/dcn_step both 5 2>&1 | grep -v "^    default vs old" >> $L
done
for t in 0 9 10; do
    echo "== dense forward / data gradient, LSNET_CONV_TILE=$t" >> $L
    LSNET_CONV_TILE=$t timeout 40 tools/ubench/conv_step 10 >> $L 2>&1
done
grep "==\|per step\|forward .* us  backward\|dcn_fwd \|dcn_wgrad \|against the host" $L
# list building: threads tap-major like the table they write (coalesced 32-byte entries, atomics spread over neighbouring anchors)
for v in 0 1; do
    echo "== deformable launches, LSNET_BIN_KMAJOR=$v" >> $L
    LSNET_BIN_KMAJOR=$v timeout 60 tools/ubench/dcn_step both 5 2>&1 | grep -v "^    default vs old" | grep -v "debug bit 28" >> $L
done
grep -A6 "LSNET_BIN_KMAJOR" $L | grep "==\|backward\|dcn_bwd_data\|against the host\|twice"
