#!/bin/bash
# round 5: deferred weight-gradient reduces -- tests that exercise the gradient arena, then the short bench with the arena limit at
# 4096 MB (one flush per step), 192 MB (partial tiles still in the Infinity Cache) and 0 (off), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_zz_grad_sink_gpu.py tests/test_resblock_gpu.py tests/test_fused_sgd_gpu.py tests/test_rccl_single_gpu.py tests/test_variants_gpu.py -q -m gpu -x \
   -k "not pose_inference" > gpurun_out/r5_c14_tests.log 2>&1; echo "tests rc $?"
tail -n 4 gpurun_out/r5_c14_tests.log
for i in 1 2; do
  for mb in 4096 192 0; do
    echo "== LSNET_WGRAD_DEFER_MB=$mb"
    LSNET_WGRAD_DEFER_MB=$mb timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
  done
done 2>&1 | tee gpurun_out/r5_c14_bench.log
