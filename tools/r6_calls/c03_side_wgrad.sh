#!/bin/bash
# round 6: weight gradients of the fused stages on the second stream (LSNET_WGRAD_SIDE=1), GPU_MAX_HW_QUEUES=2, stream-K test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "stream_k_pieces or pyramid_launch_at_bench_shape or tower_launch_at_bench_shape" > gpurun_out/r6_c03_tests.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/r6_c03_tests.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', round(d['ms_per_step'],3), 'ms', round(d['value'],2), 'img/s', {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()}, 'loss', d['loss']['loss'])"
}
for rep in 1 2 3; do
  run base A=1
  run wgrad_side LSNET_WGRAD_SIDE=1
  run queues2 GPU_MAX_HW_QUEUES=2
  run side+q2 LSNET_WGRAD_SIDE=1 GPU_MAX_HW_QUEUES=2
done > gpurun_out/r6_side_wgrad.txt 2>&1
cat gpurun_out/r6_side_wgrad.txt
