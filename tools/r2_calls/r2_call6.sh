#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -k "conv or stem" > gpurun_out/c6_pytest_conv.log 2>&1
echo "pytest conv rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c6_pytest_conv.log | tail -25
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_zz_fused_ciou_gpu.py -q --tb=short -p no:cacheprovider -k "not curve" > gpurun_out/c6_pytest_golden.log 2>&1
echo "pytest golden rc $?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/c6_pytest_golden.log | tail -10
bash tools/profile_bench.sh r2c 3 --no-extra > gpurun_out/r2c_prof.log 2>&1; head -n 30 gpurun_out/r2c_kernel_stats.txt | cut -c1-150
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/r2c_bench.log 2>&1; grep "^{" gpurun_out/r2c_bench.log | cut -c1-300
timeout 300 python tools/bench_convs_r2.py 2>&1 | grep -v amdgpu.ids | tail -34 | tee gpurun_out/r2c_convs.log | tail -12
