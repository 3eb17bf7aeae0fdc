#!/bin/bash
# first run of the two-workgroups-per-CU dense conv kernel: operator tests, then per-layer times and errors
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "conv" 2>&1 | tail -15 > gpurun_out/c01_tests.log
timeout 200 python tools/bench_convs_r2.py --own-only > gpurun_out/c01_convs.log 2>&1
tail -5 gpurun_out/c01_tests.log; cat gpurun_out/c01_convs.log
