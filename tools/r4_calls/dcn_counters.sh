#!/bin/bash
# kernel trace, then SQ counters (their own pass: --pmc never together with a trace domain other than the kernel trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
export LSNET_SO="$R/lsnet_amd/csrc/liblsnet_hip.so"
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r4_dcn_trace" -o dcn_step -- "$R/tools/ubench/dcn_step" both 3 > "$R/gpurun_out/r4_dcn_trace.log" 2>&1)
(cd /tmp && timeout 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$R/gpurun_out/r4_dcn_sq" -o dcn_step -- "$R/tools/ubench/dcn_step" tower 2 > "$R/gpurun_out/r4_dcn_sq.log" 2>&1)
ls gpurun_out/r4_dcn_trace gpurun_out/r4_dcn_sq 2>/dev/null | head
tail -5 gpurun_out/r4_dcn_trace.log
