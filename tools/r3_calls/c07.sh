#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_zz_grad_sink_gpu.py tests/test_ops_gpu.py -q -p no:cacheprovider -k "sunk or conv or group_norm or bn_ or batch_norm or dcn_pack" 2>&1 | tail -25 > gpurun_out/c07_tests.log
tail -25 gpurun_out/c07_tests.log
