#!/bin/bash
# round 6: the step with the old / new epilogue of conv_mm_kernel, alternating on one box (see c22_epilogue.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/${1:-r6_epilogue_step}.txt
: > $out
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'img/s', {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])"
}
for rep in 1 2 3; do
  run old LSNET_HIP_SO=$PWD/lsnet_amd/csrc/ab_oldepi.so
  run new LSNET_HIP_SO=$PWD/lsnet_amd/csrc/liblsnet_hip.so
done >> $out 2>&1
cat $out
