#!/bin/bash
# round 6: stream-K fix-up fetches the slots of four pieces together; partner ab_v9.so = the commit before
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv or stream_k" 2>&1 | tail -3
bash tools/r6_calls/c26_ab.sh ab_v9 r6_sk_fixup | grep -E "^==|per step|^old|^new"
