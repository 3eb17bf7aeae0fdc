#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "conv" 2>&1 | tail -3
timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids > gpurun_out/c10_convs.log
tail -32 gpurun_out/c10_convs.log | cut -c1-41,72-90,150-180
