#!/bin/bash
# A/B on one box: staging slices between the MFMAs (default build) vs in front of them (diagnostic build); step bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for so in "" lsnet_amd/csrc/liblsnet_hip_noilv.so ""; do
echo "== LSNET_HIP_SO=$so"
LSNET_HIP_SO=$so timeout 120 python tools/conv_probe.py 2>&1 | grep -v amdgpu | grep -E "P3|all5|2rounds|l3_3x3"
LSNET_HIP_SO=$so timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c21_bench.log 2>&1
grep '^{' gpurun_out/c21_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2)); print(d['roofline'])
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')" || tail -30 gpurun_out/c21_bench.log
