#!/bin/bash
# self-test of bench.py's N = 2 path on the single GPU of this box (gloo, both ranks on cuda:0), plus smoke()
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py smoke 2>&1 | tail -2
LSNET_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 3 --warmup 2 > gpurun_out/c20_bench_n2_gloo.log 2>&1
echo "rc $?"
grep '^{' gpurun_out/c20_bench_n2_gloo.log | cut -c1-2500
grep -i "error\|Traceback" -A5 gpurun_out/c20_bench_n2_gloo.log | head -30
