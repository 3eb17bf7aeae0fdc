#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "conv2d_matches" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c22_bench.log 2>&1
grep '^{' gpurun_out/c22_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2)); print(d['roofline'])
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')" || tail -30 gpurun_out/c22_bench.log
LSNET_CONV_TILE=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tile1:', round(d['value'],2), round(d['ms_per_step'],2))"
