#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LSNET_RCCL_SINGLE=1 timeout 120 python -m pytest tests/test_rccl_single_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r4_rccl_single.log
