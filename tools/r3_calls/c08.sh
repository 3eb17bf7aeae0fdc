#!/bin/bash
# new dense weight-gradient kernel (patch + transpose reads): tests, per-layer table old vs new
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_zz_grad_sink_gpu.py -q -x -p no:cacheprovider -k "conv or sunk" 2>&1 | tail -12 > gpurun_out/c08_tests.log
tail -12 gpurun_out/c08_tests.log
timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids > gpurun_out/c08_convs.log
tail -32 gpurun_out/c08_convs.log | cut -c1-150
echo "== old wgrad kernel"
LSNET_WGRAD_OLD=1 timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids | tail -32 | cut -c40-90
