#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -x -k "conv or test_dcn_forward_backward or tower_launch" > gpurun_out/c11_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c11_pytest.log | tail -5
for r in 2 1 3 4; do
  LSNET_WGRAD_ROUNDS=$r timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c11_bench_r$r.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/c11_bench_r$r.log') if x.startswith('{')]
d=json.loads(l[-1]); print('rounds $r:', round(d['value'],2), round(d['ms_per_step'],2), {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
PY
done
for t in 0 1 2 3; do
  echo "== LSNET_CONV_TILE=$t"
  LSNET_CONV_TILE=$t timeout 300 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu > gpurun_out/c11_convs_tile$t.log
  tail -1 gpurun_out/c11_convs_tile$t.log
done
