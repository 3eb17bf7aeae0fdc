#!/bin/bash
# round 5: the arena-side tests again (growth policy), then the final counters / trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_zz_grad_sink_gpu.py tests/test_variants_gpu.py tests/test_rccl_single_gpu.py -q -m gpu -x -s -k "not pose_inference" > gpurun_out/r5_c17_tests.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|library scratch" gpurun_out/r5_c17_tests.log | cut -c1-300
bash tools/r5_calls/final_counters.sh
