#!/bin/bash
# 64 x 256 tile as a default + per-anchor gather: operator tests, conv table, step bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids | tail -42 > gpurun_out/c20_convs.txt; tail -1 gpurun_out/c20_convs.txt
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2)); print(d['roofline'])
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')"
