#!/bin/bash
# gradient sinks + multi-tensor weight images: tests, then the bench and its launch census
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_zz_grad_sink_gpu.py tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "sunk or conv or group_norm or bn_ or batch_norm or dcn_pack" 2>&1 | tail -8 > gpurun_out/c06_tests.log
tail -8 gpurun_out/c06_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c06_bench.log 2>&1
grep '^{' gpurun_out/c06_bench.log | cut -c1-200 || tail -20 gpurun_out/c06_bench.log
bash tools/profile_bench.sh c06 3 --no-extra
head -30 gpurun_out/c06_kernel_stats.txt | cut -c1-180
