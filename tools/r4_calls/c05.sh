#!/bin/bash
# round 4, call 5: ATen launch sites of one step (torch profiler) + a short bench on the current tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python tools/torch_prof_sites.py > gpurun_out/r4_aten_sites.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/r4_c05_bench.log 2>&1
tail -c 1500 gpurun_out/r4_c05_bench.log
