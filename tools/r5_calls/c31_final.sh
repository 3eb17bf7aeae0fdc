#!/bin/bash
# round 5, final tree: the tests around the last changes (second stream, image rebuild guard, reducer), then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_sgd_gpu.py tests/test_golden_gpu.py tests/test_graph_gpu.py tests/test_variants_gpu.py tests/test_rccl_single_gpu.py tests/test_zz_grad_sink_gpu.py -q -m gpu -x \
  -k "sgd or curve or graph or bit_reproducible or multi_scale or iteration0 or fused_level or head_forward or rccl or deferred or one_training_step" > gpurun_out/r5_c31_tests.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r5_c31_tests.log | tail -4
timeout 1200 python bench.py > gpurun_out/r5_bench.log 2>&1
grep '^{' gpurun_out/r5_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); e=d['extra']; c=e['config3_r101_dcn_mstrain']
print('config3', c['value'], c['fixed_shape_800x1344'], c['ratio_to_fixed_shape_pixel_normalised'], c['torch_allocator_device_allocs_during_timed_passes'], c['library_hipMalloc_calls_during_timed_passes'])
print('config4', e['config4_segm_x101_dcn']['value'], 'x3', e['bf16x3']['value'], 'fp32', e['fp32']['value'], 'pose', e['infer_pose_bs4']['ms_per_batch'])
print({k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
