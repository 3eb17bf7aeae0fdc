#!/bin/bash
# s_setprio around the MFMA groups of conv_mm_kernel (variant build) against the default: conv table and step, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for so in "" lsnet_amd/csrc/liblsnet_hip_prio.so ""; do
echo "== LSNET_HIP_SO=$so"
LSNET_HIP_SO=$so timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids | tail -1
LSNET_HIP_SO=$so timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items()})"
done
