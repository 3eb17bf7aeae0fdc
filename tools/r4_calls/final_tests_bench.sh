#!/bin/bash
# round 4, final measurement call B: the full GPU suite (-s: measured deviations in the log), then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|^$\|Consider using\|got = np" > gpurun_out/r4_gpu_tests.log
tail -5 gpurun_out/r4_gpu_tests.log
timeout 900 python bench.py > gpurun_out/r4_bench.log 2>&1
tail -c 2500 gpurun_out/r4_bench.log
