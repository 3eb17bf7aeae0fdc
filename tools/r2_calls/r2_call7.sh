#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -k "bench_shape or deterministic or split6 or (test_dcn_forward_backward and (default or x3_gather or x6_atomic))" > gpurun_out/c7_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c7_pytest.log | tail -12
for cfg in "bf16x6 1" "bf16x3 1"; do
    set -- $cfg
    echo "== LSNET_MATH=$1"
    LSNET_MATH=$1 timeout 300 python tools/bench_ops.py --what dcn_all5 --iters 10 2>&1 | grep -v "^{\|amdgpu" | tee -a gpurun_out/c7_bench_ops.log
done
timeout 300 python tools/phase_clocks.py bwd1 2>&1 | grep -v amdgpu | head -9
bash tools/profile_bench.sh r2d 3 --no-extra > gpurun_out/r2d_prof.log 2>&1; head -n 16 gpurun_out/r2d_kernel_stats.txt | cut -c1-150
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/r2d_bench.log 2>&1; grep "^{" gpurun_out/r2d_bench.log | cut -c1-260
