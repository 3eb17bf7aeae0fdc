#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "conv_bn_act_folded" 2>&1 | tail -2
LSNET_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/c35_gloo2.log 2>&1
grep '^{' gpurun_out/c35_gloo2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], round(d['value'],2), round(d['ms_per_step'],1)); print({k:v for k,v in d['config'].items() if k not in ('workload','math')})" || tail -15 gpurun_out/c35_gloo2.log
