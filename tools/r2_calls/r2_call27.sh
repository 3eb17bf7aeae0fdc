#!/bin/bash
# last sanity of the default build after the final source edits (macro plumbing of the compile-time option)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
timeout 40 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "(test_conv2d_matches_torch or test_dcn_forward_backward) and (bf16x6 or default)" 2>&1 | tail -1
timeout 40 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],2), round(d['ms_per_step'],2), d['loss']['loss'])"
