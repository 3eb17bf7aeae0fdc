#!/bin/bash
# conv + trainable eval-mode BatchNorm folded into one forward launch: operator test, backbone fixtures, sink test, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_zz_grad_sink_gpu.py tests/test_graph_gpu.py tests/test_variants_gpu.py -q -p no:cacheprovider -k "conv_bn_act_folded or backbone or sunk or graph or bn_ or training_curve or one_training_step" 2>&1 | grep -E "passed|failed|FAILED|Error|low-lr|worst relative" | tail -14
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c33_bench.log 2>&1
grep '^{' gpurun_out/c33_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))
for k,v in d.get('kernels',{}).items(): print('  ',k, v['launches'], round(v['ms_per_step'],3),'ms/step', round(v['tflops'],1),'TF', round(v['alg_gbps']),'GB/s')" || tail -30 gpurun_out/c33_bench.log
