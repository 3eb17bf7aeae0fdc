#!/bin/bash
# short bench: product vs A/B library, alternating (A/B here = the library built before the change under test)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
  for so in liblsnet_hip.so liblsnet_hip_ab.so; do
    echo "== $so"
    LSNET_HIP_SO=$PWD/lsnet_amd/csrc/$so timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
  done
done 2>&1 | tee gpurun_out/r5_c09_bench.log
