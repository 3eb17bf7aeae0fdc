#!/bin/bash
# round 6: grouped deformable kernels (forward on fp32 MFMA, weight gradient on staged columns): parity, determinism, config 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "x101 or g64_c8 or g4_c16 or g16_c4 or v2_s2_g4_dg2 or v2_g2 or grouped_backward_is_deterministic" > gpurun_out/r6_c04_tests.log 2>&1; echo "tests rc $?"
tail -15 gpurun_out/r6_c04_tests.log
timeout 600 python tools/config_steps.py segm x101-dcn 3 2>&1 | grep -E "ms/step" > gpurun_out/r6_cfg4_grouped.txt
cat gpurun_out/r6_cfg4_grouped.txt
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_variants_gpu.py -q -m gpu -x -k "x101 or dcn_backbones or segm" > gpurun_out/r6_c04_tests2.log 2>&1; echo "tests2 rc $?"
tail -5 gpurun_out/r6_c04_tests2.log
