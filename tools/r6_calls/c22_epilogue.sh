#!/bin/bash
# round 6: epilogue of conv_mm_kernel fetches bias / residual / gate of all quads of a pixel tile first (buffer loads, no
# branch between them) instead of quad by quad with a full wait each.  ab_oldepi.so = the library of the commit before
# (built from `git show HEAD:lsnet_amd/csrc/conv_kernels.h` in a copy of csrc/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r6_epilogue.txt
: > $out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | tail -3 >> $out
for v in ab_oldepi liblsnet_hip ab_oldepi liblsnet_hip; do
  echo "== $v: conv_step (no epilogue operands)" >> $out
  LSNET_SO=lsnet_amd/csrc/$v.so timeout 100 tools/ubench/conv_step 10 2>&1 | tail -1 >> $out
done
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
