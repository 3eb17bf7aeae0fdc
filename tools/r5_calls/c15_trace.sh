#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_bench.sh r5c 3 --no-extra > gpurun_out/r5c_profile.log 2>&1
grep -E "reduce|wall span|wgrad" gpurun_out/r5c_kernel_stats.txt | cut -c1-200
python -c "
import sys; sys.path.insert(0,'.')
" 
