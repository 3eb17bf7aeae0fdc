#!/bin/bash
# rocprofv3 kernel trace of a short bench.py run; leaves only a small text summary of the TIMED
# steps in gpurun_out/ (the raw trace stays on the GPU box).
# usage: tools/profile_bench.sh <tag> [steps] [extra bench args]
set -u
tag=${1:-prof}; steps=${2:-3}; shift; shift || true
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
raw=/tmp/prof_$tag
rm -rf "$raw"; mkdir -p gpurun_out
LSNET_PROF_MARKERS=1 timeout 1000 rocprofv3 --kernel-trace --output-format csv -d "$raw" -o bench -- \
    python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/${tag}_run.log 2>&1
grep '^{' gpurun_out/${tag}_run.log | cut -c1-300
python tools/prof_summary.py "$raw" gpurun_out/${tag}_kernel_stats.txt $steps
