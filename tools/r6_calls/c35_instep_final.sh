#!/bin/bash
# round 6, end: the in-step vs isolated table of the dense convolutions again, on the final kernels (epilogue operands fetched first)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/instep_vs_isolated.py 3 > gpurun_out/r6_instep_vs_isolated_final.txt 2> gpurun_out/r6_instep_err.log; echo "instep rc $?"
tail -4 gpurun_out/r6_instep_vs_isolated_final.txt
bash tools/r6_calls/c16_cfg_traces.sh > /dev/null 2>&1
head -12 gpurun_out/r6_cfg4_final_run.log | cut -c1-200
