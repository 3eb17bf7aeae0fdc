#!/bin/bash
# HBM counters of the step's HBM-BOUND kernel classes (GroupNorm, focal loss, cross-IOU rows, top-k, offset
# chain, ReLU gates, clip + SGD, NMS): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md) over two
# training steps of the benchmark model, restricted to those kernels, plus a --kernel-trace pass for their durations.
# Writes gpurun_out/<tag>_pmc_hbm_stream.txt.  Hard limit per pass.
set -u
tag=${1:-r5}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
RE='lsn::(gn_|focal_|cross_iou|ciou|topk_|nms|relu_gate|sgd_|chain_|offset_chain|bn_act|bn_param|image_prep|permute)'
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-timing"
for ctr in FETCH_SIZE WRITE_SIZE; do
    raw=/tmp/pmcstream_${tag}_$ctr
    rm -rf "$raw"
    LSNET_PROF_MARKERS=1 timeout -s KILL 300 rocprofv3 --pmc $ctr --kernel-include-regex "$RE|selftest32" --output-format csv -d "$raw" -o ops -- \
        $CMD > gpurun_out/${tag}_pmcstream_${ctr}_run.log 2>&1
    echo "pass $ctr exit $?"
done
raw=/tmp/pmcstream_${tag}_trace
rm -rf "$raw"
LSNET_PROF_MARKERS=1 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$raw" -o ops -- $CMD > gpurun_out/${tag}_pmcstream_trace_run.log 2>&1
echo "pass trace exit $?"
python tools/pmc_stream_summary.py /tmp/pmcstream_${tag}_FETCH_SIZE /tmp/pmcstream_${tag}_WRITE_SIZE /tmp/pmcstream_${tag}_trace 2 \
    > gpurun_out/${tag}_pmc_hbm_stream.txt 2>&1
cat gpurun_out/${tag}_pmc_hbm_stream.txt
