#!/bin/bash
# round 6: the 256-channel blocks of the per-anchor sums as blockIdx.y for C > 256 (config 4's backbone layers); debug bit 16 = 65536:
# one wave walks all blocks, as before
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "dcn_forward_backward or grouped or deterministic or tower_launch or pyramid_launch" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_variants_gpu.py -q -m gpu -x -k "x101 or dcn_backbones or res2net" 2>&1 | tail -3
out=gpurun_out/r6_anchor_cb.txt
: > $out
for rep in 1 2; do
  for dbg in 65536 0; do
    echo "== debug word $dbg (rep $rep)" >> $out
    CONFIG_STEPS_DBG=$dbg timeout 600 python tools/config_steps.py segm x101-dcn 3 2>&1 | grep -E "ms/step" | head -8 >> $out
  done
done
cat $out
