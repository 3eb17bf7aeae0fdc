#!/bin/bash
# round 6: kernel trace of the headline step on the current tree + the GPU tests touched so far
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/profile_bench.sh r6 3 --no-extra > gpurun_out/r6_profile_stdout.log 2>&1; echo "profile rc $?"
head -60 gpurun_out/r6_kernel_stats.txt | cut -c1-200
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_rccl_single_gpu.py tests/test_resblock_gpu.py -q -m gpu -x > gpurun_out/r6_c05_tests.log 2>&1; echo "tests rc $?"
tail -5 gpurun_out/r6_c05_tests.log
