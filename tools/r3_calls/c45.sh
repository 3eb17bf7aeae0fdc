#!/bin/bash
# end-to-end sanity of the final tree: a short default bench (loss against profiles/r3_bench.log: 3.52975)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 24 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c45_bench.log 2>&1
echo "rc $?"
grep '^{' gpurun_out/c45_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), d['loss'], {k:round(v['ms_per_step'],2) for k,v in d.get('kernels',{}).items()})" || tail -3 gpurun_out/c45_bench.log
