#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -k "conv or stem" > gpurun_out/c5_pytest_conv.log 2>&1
echo "pytest conv rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/c5_pytest_conv.log | tail -25
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_variants_gpu.py tests/test_zz_fused_ciou_gpu.py -q --tb=short -p no:cacheprovider -s > gpurun_out/c5_pytest_golden.log 2>&1
echo "pytest golden rc $?"; grep -E "passed|failed|^FAILED|^ERROR|worst" gpurun_out/c5_pytest_golden.log | tail -30
bash tools/profile_bench.sh r2b 3 --no-extra > gpurun_out/r2b_prof.log 2>&1; head -n 40 gpurun_out/r2b_kernel_stats.txt; grep -c "igemm\|ck::" gpurun_out/r2b_kernel_stats.txt
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/r2b_bench.log 2>&1; grep "^{" gpurun_out/r2b_bench.log | cut -c1-300
