#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_variants_gpu.py tests/test_graph_gpu.py -q -m gpu -x \
  -k "group_norm or head_forward or head_at_256 or bit_reproducible or iteration0 or graph or decode_is or residual_in_the_epilogue or multi_level" > gpurun_out/r5_c18_tests.log 2>&1; echo "tests rc $?"
tail -n 3 gpurun_out/r5_c18_tests.log
timeout 60 tools/ubench/norm_step 5 | tail -4
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
done
