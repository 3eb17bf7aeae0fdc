#!/bin/bash
# round 5, call 5: the round's new / touched GPU tests, then the short bench (no CPU baseline, no extra legs) on the product
# library and on the A/B library (= round 4's work distribution), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv2d or conv_bn or stream_k or split6 or module_dispatch" > gpurun_out/r5_c03_tests_conv.log 2>&1; echo "conv tests rc $?"
tail -n 3 gpurun_out/r5_c03_tests_conv.log
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_resblock_gpu.py tests/test_variants_gpu.py -q -m gpu -x -s -k "decode_is_exact or multiscale or resblock or iteration0 or bit_reproducible" > gpurun_out/r5_c03_tests_misc.log 2>&1; echo "misc tests rc $?"
tail -n 5 gpurun_out/r5_c03_tests_misc.log
for i in 1 2; do
  for so in liblsnet_hip.so liblsnet_hip_ab.so; do
    echo "== $so"
    LSNET_HIP_SO=$PWD/lsnet_amd/csrc/$so timeout 600 python bench.py --no-cpu-baseline --no-extra 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
  done
done 2>&1 | tee gpurun_out/r5_c03_bench_ab.log
