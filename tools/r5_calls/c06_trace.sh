#!/bin/bash
# kernel trace of the timed steps of the short bench (3 steps), with the idle-gap table of tools/prof_summary.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/profile_bench.sh r5a 3 --no-extra > gpurun_out/r5a_profile.log 2>&1
tail -n 60 gpurun_out/r5a_kernel_stats.txt
