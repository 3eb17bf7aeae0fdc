#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for parts in 0 1 2 4; do
echo "== LSNET_DCN_MM_PARTS=$parts"
LSNET_DCN_MM_PARTS=$parts timeout 300 python -m pytest tests/test_golden_gpu.py -q -x -p no:cacheprovider -k "256" 2>&1 | grep -E "passed|failed|AssertionError" | head -4
done
timeout 300 python -m pytest tests/test_graph_gpu.py tests/test_runner_gpu.py -q -p no:cacheprovider -k "graph" 2>&1 | tail -3
