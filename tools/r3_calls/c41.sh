#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 200 python -m pytest tests/test_golden_gpu.py -q -p no:cacheprovider -k "low_learning_rate" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/r3_bench.log 2>&1
grep '^{' gpurun_out/r3_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2)); r=d['roofline']; print(r['family'], round(r['achieved'],1), round(r['frac'],3), r['traffic']); print(d['cpu_baseline']['value']); print({k:(round(v.get('value',v.get('img_per_s',0)),1)) for k,v in d.get('extra',{}).items()})
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')" || tail -5 gpurun_out/r3_bench.log
