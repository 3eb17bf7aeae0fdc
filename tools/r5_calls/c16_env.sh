#!/bin/bash
# round 5: HIP runtime switches that change the per-launch cost, short bench each, twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'loss', d['loss']['loss'])" || tail -3 gpurun_out/bench_err.log
}
for i in 1 2; do
  run X=1
  run HIP_FORCE_DEV_KERNARG=1
  run HIP_FORCE_DEV_KERNARG=0
  run GPU_MAX_HW_QUEUES=2
  run HSA_NO_SCRATCH_RECLAIM=1
  run HIP_FORCE_DEV_KERNARG=1 HSA_NO_SCRATCH_RECLAIM=1 GPU_MAX_HW_QUEUES=2
done 2>&1 | tee gpurun_out/r5_c16_env.log
