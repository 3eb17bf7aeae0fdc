#!/bin/bash
# round 6: column gradients on the fp32 matrix instructions (grouped calls in every mode, dense calls of the exact mode)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "x101 or g64 or g4_c16 or g16_c4 or math_fp32 or exact_mode or deterministic or v2_g2 or g4_dg2" > gpurun_out/r6_c11_tests.log 2>&1; echo "tests rc $?"
tail -4 gpurun_out/r6_c11_tests.log
timeout 600 python tools/config_steps.py segm x101-dcn 3 2>&1 | grep -E "ms/step" > gpurun_out/r6_cfg4_gcol_mfma.txt
cat gpurun_out/r6_cfg4_gcol_mfma.txt
for m in fp32 bf16x6; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --math $m 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$m', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'img/s', {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])"
done > gpurun_out/r6_exact_mode_step.txt 2>&1
cat gpurun_out/r6_exact_mode_step.txt
