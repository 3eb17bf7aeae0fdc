#!/bin/bash
# round 6: deformable weight images from the per-optimizer-step cache (LSNET_CACHE_DCN_IMAGES=0: in-call build, as until round 5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_fused_sgd_gpu.py -q -m gpu -x -k "weight_images or graph or sgd or tower_launch_at_bench_shape or pyramid_launch" > gpurun_out/r6_c12_tests.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/r6_c12_tests.log
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'img/s', {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])"
}
for rep in 1 2 3; do
  run in_call LSNET_CACHE_DCN_IMAGES=0
  run cached LSNET_CACHE_DCN_IMAGES=1
done > gpurun_out/r6_dcn_images.txt 2>&1
cat gpurun_out/r6_dcn_images.txt
