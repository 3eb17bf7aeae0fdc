#!/bin/bash
# round 5: the R-101-DCN step (BASELINE config 3, fixed shape) lost 19 % between the mid-round and the final full bench: which switch?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" timeout 300 python bench.py --backbone r101-dcn --no-cpu-baseline --no-extra --steps 6 --warmup 3 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
}
{
run X=1
run LSNET_SIDE_STREAM_IMAGES=0
run LSNET_SIDE_STREAM_TARGETS=0
run LSNET_FUSED_LEVEL_SUMS=0
run LSNET_GATE_MULTI=0
run LSNET_WGRAD_DEFER_MB=0
run LSNET_SIDE_STREAM_IMAGES=0 LSNET_SIDE_STREAM_TARGETS=0
run X=1
} 2>&1 | tee gpurun_out/r5_c27_r101.log
