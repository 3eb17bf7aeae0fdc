#!/bin/bash
# new tests (256-channel head, full-size configs 3/4/5, dispatch, bench-shape determinism), HBM counters of the DCN
# launches of the step, the 2-rank gloo self-test of bench.py's N > 1 path
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_golden_gpu.py tests/test_variants_gpu.py tests/test_ops_gpu.py -q -p no:cacheprovider -s -k "256 or full_size or dispatch or tower_launch or one_training_step" 2>&1 | grep -v "amdgpu\|Warn\|warn\|got = " > gpurun_out/c16_tests.log
grep -E "same device|passed|failed|FAILED|Error" gpurun_out/c16_tests.log | tail -14
bash tools/pmc_step_shapes.sh r3 2>&1 | tail -12
LSNET_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/c16_gloo2.log 2>&1
grep '^{' gpurun_out/c16_gloo2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], round(d['value'],2), round(d['ms_per_step'],1)); print({k:v for k,v in d['config'].items() if k not in ('workload','math')})" || tail -15 gpurun_out/c16_gloo2.log
