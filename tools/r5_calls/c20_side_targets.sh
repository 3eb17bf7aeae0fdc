#!/bin/bash
# round 5: init-stage targets on a second stream, in the shadow of the head's forward (LSNET_SIDE_STREAM_TARGETS=0 = as before):
# head fixtures, graph replay, multi-scale, reproducibility, then the short bench alternating on this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_variants_gpu.py tests/test_graph_gpu.py tests/test_zz_grad_sink_gpu.py -q -m gpu -x \
  -k "head_forward or head_at_256 or bit_reproducible or iteration0 or graph or multi_scale or curve or deferred" > gpurun_out/r5_c20_tests.log 2>&1; echo "tests rc $?"
tail -n 4 gpurun_out/r5_c20_tests.log
for sw in 0 1 0 1; do
  echo "== LSNET_SIDE_STREAM_TARGETS=$sw"
  LSNET_SIDE_STREAM_TARGETS=$sw timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
done 2>&1 | tee gpurun_out/r5_c20_bench.log
