#!/bin/bash
# SQ counters (matrix-pipe busy cycles, wave cycles, stalls, LDS bank conflicts) of the deformable-conv launches of one
# benchmark step, replayed by tools/step_shapes.py.  Counter passes only (no trace domains); hard 120 s limit per pass.
# Writes gpurun_out/<tag>_pmc_sq.txt.  usage: tools/pmc_sq_step_shapes.sh <tag>
set -u
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/${tag}_pmc_sq.txt
: > "$out"
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
    i=$((i + 1))
    raw=/tmp/pmcsq_${tag}_$i
    rm -rf "$raw"
    timeout -s KILL 120 rocprofv3 --pmc $set --kernel-include-regex 'lsn::' --output-format csv -d "$raw" -o ops -- \
        python tools/step_shapes.py > gpurun_out/${tag}_pmc_sq_run$i.log 2>&1
    echo "pass $i exit $?" >> "$out"
    python tools/pmc_summary.py "$raw" >> "$out" 2>&1
done
cat "$out" | head -150
