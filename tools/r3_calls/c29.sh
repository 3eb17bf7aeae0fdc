#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_zz_grad_sink_gpu.py -q -p no:cacheprovider -s -k "training_curve or 256 or (x3_windowed and (wide or l4)) or graph or sunk" 2>&1 | grep -v "amdgpu\|Warn\|warn\|got = " > gpurun_out/c29_tests.log
grep -E "^curve|low-lr|worst relative|same device|passed|failed|FAILED|Error" gpurun_out/c29_tests.log | tail -40
