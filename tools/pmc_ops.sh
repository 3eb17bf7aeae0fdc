#!/bin/bash
# HBM counters (FETCH_SIZE, WRITE_SIZE: separate passes) of the deformable-conv kernels on the all-5-level tower shape,
# from the micro-benchmark instead of the whole step (a counter pass over bench.py hung for 17 minutes in round 1).
# Hard 90 s limit per pass.  usage: tools/pmc_ops.sh <tag>
set -u
tag=${1:-pmcops}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
    raw=/tmp/pmcops_${tag}_$ctr
    rm -rf "$raw"
    timeout -s KILL 90 rocprofv3 --pmc $ctr --output-format csv -d "$raw" -o ops -- \
        python tools/bench_ops.py --what dcn_all5 --iters 4 > gpurun_out/${tag}_${ctr}_run.log 2>&1
    echo "pass $ctr exit $?"
    python tools/pmc_summary.py "$raw" > gpurun_out/${tag}_${ctr}.txt 2>&1
done
cat gpurun_out/${tag}_FETCH_SIZE.txt gpurun_out/${tag}_WRITE_SIZE.txt
