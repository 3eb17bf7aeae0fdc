#!/bin/bash
# round 6: stream-K pieces for the small (backbone) launches of the deformable forward: config 3 with / without (debug bit 19 is
# not reachable from here: two builds would be needed; the family table of config_steps says it directly)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python tools/config_steps.py bbox r101-dcn 3 2>&1 | grep -E "ms/step" > gpurun_out/r6_cfg3_sk_small.txt
cat gpurun_out/r6_cfg3_sk_small.txt
timeout 600 python tools/config_steps.py segm x101-dcn 3 2>&1 | grep -E "ms/step" > gpurun_out/r6_cfg4_sk_small.txt
cat gpurun_out/r6_cfg4_sk_small.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "stream_k_pieces or r101 or v2_head_p6 or pyr_head or tower_launch_at_bench_shape" > gpurun_out/r6_c07_tests.log 2>&1; echo "tests rc $?"
tail -3 gpurun_out/r6_c07_tests.log
