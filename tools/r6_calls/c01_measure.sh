#!/bin/bash
# round 6, first call: measurements only on the tree as round 5 left it (+ the per-call log)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/instep_vs_isolated.py 3 > gpurun_out/r6_instep_vs_isolated.txt 2> gpurun_out/r6_instep_err.log; echo "instep rc $?"
tail -4 gpurun_out/r6_instep_vs_isolated.txt
for cfg in "segm x101-dcn cfg4" "bbox r101-dcn cfg3"; do
  set -- $cfg
  raw=/tmp/prof_$3; rm -rf $raw
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $raw -o t -- python tools/config_steps.py $1 $2 3 > gpurun_out/r6_$3_run.log 2>&1
  echo "$3 rc $?"; grep -E "ms/step" gpurun_out/r6_$3_run.log | head -12
  python tools/prof_summary.py $raw gpurun_out/r6_$3_kernel_stats.txt 3 > /dev/null
  head -30 gpurun_out/r6_$3_kernel_stats.txt
done
timeout 1500 python tools/rccl_streams.py > gpurun_out/r6_rccl_streams.txt 2>&1; echo "streams rc $?"
cat gpurun_out/r6_rccl_streams.txt
