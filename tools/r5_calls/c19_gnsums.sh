#!/bin/bash
# round 5: GroupNorm statistics sums in two alternating library buffers (no memset launch): operator tests, then the short bench with
# the committed library (liblsnet_hip_ab.so, built from the parent commit) and the new one alternating on this box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_variants_gpu.py -q -m gpu -x \
  -k "group_norm or bit_reproducible or multi_scale" > gpurun_out/r5_c19_tests.log 2>&1; echo "tests rc $?"
tail -n 4 gpurun_out/r5_c19_tests.log
for so in liblsnet_hip_ab.so liblsnet_hip.so liblsnet_hip_ab.so liblsnet_hip.so; do
  echo "== $so"
  LSNET_HIP_SO=$PWD/lsnet_amd/csrc/$so timeout 600 python bench.py --no-cpu-baseline --no-extra 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
done 2>&1 | tee gpurun_out/r5_c19_bench.log
