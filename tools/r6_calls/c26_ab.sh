#!/bin/bash
# round 6: generic A/B of two builds of the library: c26_ab.sh <partner .so name> <tag>  -- harnesses, then the step alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
old=${1:-ab_v4}; tag=${2:-r6_ab}
out=gpurun_out/$tag.txt
: > $out
for rep in 1 2; do
  for v in $old liblsnet_hip; do
    echo "== $v (rep $rep)" >> $out
    LSNET_SO=lsnet_amd/csrc/$v.so timeout 100 tools/ubench/conv_step 10 2>&1 | tail -1 >> $out
    timeout 100 tools/ubench/wgrad_ab lsnet_amd/csrc/$v.so rule 2>&1 | tail -1 | sed 's/^/wgrad_ab rule (old = patch kernel everywhere, new = as routed): /' >> $out
    LSNET_SO=lsnet_amd/csrc/$v.so timeout 100 tools/ubench/dcn_step both 10 2>&1 | grep -v "debug bit 28" | grep -E "backward twice|^    dcn_" | awk '/backward twice/{stop=1} !stop{print} /pyramid/{stop=0}' | head -8 >> $out
  done
done
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'img/s', {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])"
}
for rep in 1 2 3; do
  run old LSNET_HIP_SO=$PWD/lsnet_amd/csrc/$old.so
  run new LSNET_HIP_SO=$PWD/lsnet_amd/csrc/liblsnet_hip.so
done >> $out 2>&1
cat $out
