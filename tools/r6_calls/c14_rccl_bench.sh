#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rccl_single_gpu.py -q -m gpu -s 2>&1 | grep -E "passed|failed|step with|Error" | tail -5
bash tools/r6_calls/full_bench.sh | tail -c 200
