#!/bin/bash
# round 5: the ReLU gates of a multi-level convolution in one launch, sliced gradients read where they lie (no copy): operator tests,
# head fixtures, reproducibility; then the short bench, parent commit's library + Python (git stash is not available on the box:
# LSNET_GATE_MULTI=0 takes the per-level path) against the new one
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_variants_gpu.py -q -m gpu -x \
  -k "relu_gate or conv2d_multi or head_forward or head_at_256 or bit_reproducible or fused_level or iteration0" > gpurun_out/r5_c25_tests.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r5_c25_tests.log | tail -5
for sw in 0 1 0 1; do
  echo "== LSNET_GATE_MULTI=$sw"
  LSNET_GATE_MULTI=$sw timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
done 2>&1 | tee gpurun_out/r5_c25_bench.log
