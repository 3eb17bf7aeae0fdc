#!/bin/bash
# round-2 measurement pass: HBM counters over the step's deformable-conv launch shapes, the default bench line
# (CPU baseline included), a kernel trace of the timed steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/pmc_step_shapes.sh r2 > gpurun_out/r2_pmc_run.txt 2>&1
tail -4 gpurun_out/r2_pmc_run.txt
[ -s gpurun_out/r2_hbm_traffic.json ] && cp gpurun_out/r2_hbm_traffic.json profiles/r2_hbm_traffic.json
s=$(date +%s)
timeout 1200 python bench.py > gpurun_out/r2_bench.log 2>&1
echo "bench rc $? in $(( $(date +%s) - s )) s"
grep '^{' gpurun_out/r2_bench.log | cut -c1-1200
bash tools/profile_bench.sh r2h 3 > gpurun_out/r2h_prof.log 2>&1
head -24 gpurun_out/r2h_kernel_stats.txt | cut -c1-150
bash tools/pmc_sq_step_shapes.sh r2 > gpurun_out/r2_pmc_sq_run.txt 2>&1
grep -c . gpurun_out/r2_pmc_sq.txt
