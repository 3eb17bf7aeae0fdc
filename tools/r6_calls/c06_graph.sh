#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for q in 4 2; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python -m pytest tests/test_graph_gpu.py -q -m gpu -x > gpurun_out/r6_c06_graph_q$q.log 2>&1; echo "queues $q: rc $?"
  tail -3 gpurun_out/r6_c06_graph_q$q.log | cut -c1-200
done
