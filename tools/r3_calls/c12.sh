#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_zz_grad_sink_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/c12_tests.log
tail -4 gpurun_out/c12_tests.log
timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids > gpurun_out/c12_convs.log
tail -1 gpurun_out/c12_convs.log
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c12_bench.log 2>&1
grep '^{' gpurun_out/c12_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2)); 
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['total_ms']/10,3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')
print(d.get('roofline'))"
