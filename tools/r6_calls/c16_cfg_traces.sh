#!/bin/bash
# round 6, final tree: kernel traces of configs 4 and 3 (the "after" of profiles/r6_cfg{3,4}_kernel_stats.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "segm x101-dcn cfg4" "bbox r101-dcn cfg3"; do
  set -- $cfg
  raw=/tmp/prof_$3; rm -rf $raw
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $raw -o t -- python tools/config_steps.py $1 $2 3 > gpurun_out/r6_$3_final_run.log 2>&1
  echo "$3 rc $?"; grep -E "ms/step" gpurun_out/r6_$3_final_run.log | head -9
  python tools/prof_summary.py $raw gpurun_out/r6_$3_kernel_stats_final.txt 3 > /dev/null
  head -24 gpurun_out/r6_$3_kernel_stats_final.txt | cut -c1-170
done
