#!/bin/bash
# round 5: the stale weight images rebuilt on a second stream behind the optimizer step's event (LSNET_SIDE_STREAM_IMAGES=0 = in stream
# order, when the first trainable convolution asks): optimizer / runner / curve / graph tests, then the short bench alternating on this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_sgd_gpu.py tests/test_golden_gpu.py tests/test_graph_gpu.py tests/test_variants_gpu.py tests/test_ops_gpu.py -q -m gpu -x \
  -k "sgd or curve or graph or bit_reproducible or multi_scale or weight_image or prepared or module_dispatch or iteration0" > gpurun_out/r5_c24_tests.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r5_c24_tests.log | tail -5
for sw in 0 1 0 1; do
  echo "== LSNET_SIDE_STREAM_IMAGES=$sw"
  LSNET_SIDE_STREAM_IMAGES=$sw timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
done 2>&1 | tee gpurun_out/r5_c24_bench.log
