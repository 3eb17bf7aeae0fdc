#!/bin/bash
# frozen conv+BN folding, fused SGD, survey/dominant bench measurement: golden + variants tests, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_golden_gpu.py tests/test_zz_grad_sink_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/c14_tests.log
tail -6 gpurun_out/c14_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c14_bench.log 2>&1
grep '^{' gpurun_out/c14_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), d['loss'])
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')
r=d.get('roofline'); print({k:r[k] for k in ('family','bound','achieved','peak','frac','launches_timed','avg_launch_ms','ms_per_step')})" || tail -20 gpurun_out/c14_bench.log
