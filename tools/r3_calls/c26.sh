#!/bin/bash
# step bench with the dcn_mm forward + unweighted backward GEMM / per-anchor pipeline, against the previous kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for mm in 1 0; do
LSNET_DCN_MM=$mm timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c26_bench_$mm.log 2>&1
grep '^{' gpurun_out/c26_bench_$mm.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('MM=$mm', round(d['value'],2), round(d['ms_per_step'],2)); r=d['roofline']; print('  ', r['family'], round(r['achieved'],1), round(r['frac'],3))
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')" || tail -30 gpurun_out/c26_bench_$mm.log
done
