#!/bin/bash
# last verification of the round: full -m gpu suite, smoke, short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -v "amdgpu\|UserWarning\|warnings.warn\|got = " > gpurun_out/r3_gpu_tests.log
grep -E " passed|failed|FAILED|Error" gpurun_out/r3_gpu_tests.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items()})"
