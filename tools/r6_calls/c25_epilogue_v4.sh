#!/bin/bash
# round 6: the unified epilogue (rows that are not whole quads -- the 27 offset / mask channels -- fetch their operands first
# too) against the library of the commit before the epilogue work (ab_oldepi.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_resblock_gpu.py -q -m gpu -x -k "conv or resblock or bottleneck" 2>&1 | tail -3
bash tools/r6_calls/c23_conv_step_ab.sh
grep "offset conv" gpurun_out/cs_ab_oldepi.txt gpurun_out/cs_liblsnet_hip.txt
bash tools/r6_calls/c24_epilogue_step.sh r6_epilogue_step_v4
