#!/bin/bash
# round 5: every loss term's levels in one launch (LSNET_FUSED_LEVEL_SUMS=0 = level by level as before): operator + step
# equivalence tests, head fixtures, then the short bench alternating on this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_variants_gpu.py tests/test_golden_gpu.py tests/test_graph_gpu.py -q -m gpu -x -s \
  -k "per_level or fused_level or focal or runner or head_forward or head_at_256 or bit_reproducible or iteration0 or graph or curve" > gpurun_out/r5_c22_tests.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|worst parameter|Error" gpurun_out/r5_c22_tests.log | tail -8
for sw in 0 1 0 1; do
  echo "== LSNET_FUSED_LEVEL_SUMS=$sw"
  LSNET_FUSED_LEVEL_SUMS=$sw timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss'], 'ref', d.get('loss_ref_rel_err'))" || tail -5 gpurun_out/bench_err.log
done 2>&1 | tee gpurun_out/r5_c22_bench.log
