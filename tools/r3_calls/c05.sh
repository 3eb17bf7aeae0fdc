#!/bin/bash
# whole step with the new dense-conv kernels: bench line + kernel trace of the timed steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c05_bench.log 2>&1
grep '^{' gpurun_out/c05_bench.log | cut -c1-1500
bash tools/profile_bench.sh c05 3 --no-extra
head -45 gpurun_out/c05_kernel_stats.txt | cut -c1-200
