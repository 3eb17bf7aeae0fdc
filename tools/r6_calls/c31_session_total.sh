#!/bin/bash
# round 6: the library of the round's first-half end (commit 048a554, ab_start.so: `git archive 048a554 lsnet_amd/csrc include`, built
# there) against the final library, the step alternating four times on ONE box: the sum of the second half's kernel changes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r6_second_half_total.txt
: > $out
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'img/s', {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])"
}
for rep in 1 2 3 4; do
  run first_half_end LSNET_HIP_SO=$PWD/lsnet_amd/csrc/ab_start.so
  run final LSNET_HIP_SO=$PWD/lsnet_amd/csrc/liblsnet_hip.so
done >> $out 2>&1
cat $out
