#!/bin/bash
# round 4, call 3: full GPU suite on the new defaults + fused bottleneck backward + RCCL tests, then a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r4_c03_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/r4_c03_bench.log 2>&1
tail -40 gpurun_out/r4_c03_tests.log; tail -c 3000 gpurun_out/r4_c03_bench.log
