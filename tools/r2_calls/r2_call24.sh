#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/phase_clocks.py conv 2>&1 | grep -v amdgpu | grep "conv forward\|mfma done / post\|loads issued / mfma" | head -16
timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -x -k "conv or stem" > gpurun_out/c24_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c24_pytest.log | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c24_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c24_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), d['loss'])
PY
