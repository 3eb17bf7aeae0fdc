#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for so in liblsnet_hip.so liblsnet_hip_gu8.so; do
  LSNET_HIP_SO=$PWD/lsnet_amd/csrc/$so timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c21_bench.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/c21_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print('$so:', round(d['value'],2), round(d['ms_per_step'],2), {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
PY
done
LSNET_HIP_SO=$PWD/lsnet_amd/csrc/liblsnet_hip_gu8.so timeout 300 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "pyramid_launch_at_bench_shape and default" 2>&1 | tail -2
