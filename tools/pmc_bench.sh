#!/bin/bash
# HBM-traffic counters for a short bench.py run: FETCH_SIZE and WRITE_SIZE need separate passes
# (TCC has 4 slots; MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counter runs use --pmc only.
# usage: tools/pmc_bench.sh <tag> [steps]
set -u
tag=${1:-pmc}; steps=${2:-2}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
    raw=/tmp/pmc_${tag}_$ctr
    rm -rf "$raw"
    timeout 1000 rocprofv3 --pmc $ctr --output-format csv -d "$raw" -o bench -- \
        python bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/${tag}_${ctr}_run.log 2>&1
    python tools/pmc_summary.py "$raw" > gpurun_out/${tag}_${ctr}.txt 2>&1
done
cat gpurun_out/${tag}_FETCH_SIZE.txt gpurun_out/${tag}_WRITE_SIZE.txt
