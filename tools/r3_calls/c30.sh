#!/bin/bash
# final measurements of the round: HBM counters of the deformable launches, default bench line (with the CPU baseline)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/pmc_step_shapes.sh r3 2>&1 | tail -30
cp gpurun_out/r3_hbm_traffic.json profiles/r3_hbm_traffic.json 2>/dev/null
timeout 900 python bench.py > gpurun_out/r3_bench.log 2>&1
grep '^{' gpurun_out/r3_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2)); print(d['roofline']); print(d['cpu_baseline']); print({k:v for k,v in d.get('extra',{}).items()})" || tail -20 gpurun_out/r3_bench.log
