#!/bin/bash
# round-2 GPU call 1: parity of the new arithmetic / backward path, then kernel timings per mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_zz_fused_ciou_gpu.py -q --tb=short -p no:cacheprovider -s \
    > gpurun_out/c1_pytest_ops.log 2>&1
echo "pytest ops rc $?"
tail -n 40 gpurun_out/c1_pytest_ops.log
for cfg in "bf16x6 1" "bf16x6 0" "bf16x3 1" "bf16x3 0"; do
    set -- $cfg
    echo "== LSNET_MATH=$1 LSNET_BWD_GATHER=$2"
    LSNET_MATH=$1 LSNET_BWD_GATHER=$2 timeout 300 python tools/bench_ops.py --what dcn_all5 --iters 10 2>&1 | grep -v "^{" | tee -a gpurun_out/c1_bench_ops.log
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c1_bench.log 2>&1
echo "bench rc $?"
tail -c 3000 gpurun_out/c1_bench.log
