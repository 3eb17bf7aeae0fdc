#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -k "bench_shape or deterministic or split6 or test_dcn_forward_backward or multi_level_launch or named_entry or fused_offset or conv" > gpurun_out/c9_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c9_pytest.log | tail -12
timeout 300 python tools/phase_clocks.py wg 2>&1 | grep -v amdgpu | head -9
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/r2f_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r2f_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
PY
