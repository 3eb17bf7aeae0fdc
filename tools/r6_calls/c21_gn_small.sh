#!/bin/bash
# round 6: GroupNorm with one or two channels per group on norm.hip; the golden head / CPV tests must not warn about an
# ATen fallback outside their deliberate same-device arm (-W error)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "group_norm" 2>&1 | tail -5 > gpurun_out/r6_gn_small.log
timeout 1500 python -m pytest tests/test_golden_gpu.py tests/test_zz_cpv_gpu.py -q -m gpu -W error::RuntimeWarning 2>&1 | tail -25 >> gpurun_out/r6_gn_small.log
cat gpurun_out/r6_gn_small.log
