#!/bin/bash
# ordered GroupNorm / focal sums, long anchor lists sorted: operator tests, reproducibility of a step, kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -q -p no:cacheprovider -k "group_norm or focal or deterministic or pyramid_launch or head_forward" 2>&1 | tail -3
timeout 300 python tools/repro_step.py 2>&1 | grep -E "^loss|parameter gradients differ|^  [a-z]" | head -24
timeout 300 python tools/fuzz_dcn.py 6 3 2>&1 | grep -E "FAIL|worst"
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items()})"
