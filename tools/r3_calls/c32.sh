#!/bin/bash
# final verification of the round: full -m gpu suite, smoke()
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -v "amdgpu\|UserWarning\|warnings.warn\|got = " > gpurun_out/r3_gpu_tests.log
grep -E " passed|failed|FAILED|Error" gpurun_out/r3_gpu_tests.log | tail -15
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
