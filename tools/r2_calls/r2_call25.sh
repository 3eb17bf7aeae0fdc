#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -x -k "test_dcn_forward_backward or bench_shape or split6 or named_entry or fused_offset or multi_level_launch" > gpurun_out/c25_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c25_pytest.log | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c25_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c25_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
PY
