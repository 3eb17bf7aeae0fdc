#!/bin/bash
# final artifacts of the round: default bench line, kernel census, full -m gpu suite, smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python bench.py > gpurun_out/r3_bench.log 2>&1
grep '^{' gpurun_out/r3_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2)); r=d['roofline']; print(r['family'], round(r['achieved'],1), round(r['frac'],3), r['traffic']); print(d['cpu_baseline']['value']); print({k:(round(v.get('value',v.get('img_per_s',0)),1)) for k,v in d.get('extra',{}).items()})
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')" || tail -20 gpurun_out/r3_bench.log
bash tools/profile_bench.sh r3d 3 --no-extra > /dev/null 2>&1
head -8 gpurun_out/r3d_kernel_stats.txt | cut -c1-150
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 | grep -v "amdgpu\|UserWarning\|warnings.warn\|got = " > gpurun_out/r3_gpu_tests.log
grep -E " passed|failed|FAILED|Error" gpurun_out/r3_gpu_tests.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
