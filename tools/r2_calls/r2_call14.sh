#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -x -k "bench_shape or deterministic or test_dcn_forward_backward or multi_level_launch or named_entry or fused_offset" > gpurun_out/c14_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c14_pytest.log | tail -5
bash tools/profile_bench.sh r2i 3 > gpurun_out/r2i_prof.log 2>&1
head -16 gpurun_out/r2i_kernel_stats.txt | cut -c1-140
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c14_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c14_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
PY
