#!/bin/bash
# round 6: smoke() and the full GPU suite (-s: measured deviations in the log)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q -s --tb=short -p no:cacheprovider 2>&1 | grep -v "Warning\|warnings.warn\|^$\|Consider using\|got = np" > gpurun_out/r6_gpu_tests.log
tail -5 gpurun_out/r6_gpu_tests.log
