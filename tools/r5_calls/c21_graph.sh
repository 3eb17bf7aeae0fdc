#!/bin/bash
# round 5: how much of the eager step is still the CPU's launch rate?  The same step replayed from a hipGraph (no launch cost at all)
# beside the eager one, alternating on this box; then the sections of the eager step (events, no tracer)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for mode in "" "--graph" "" "--graph"; do
  echo "== bench.py $mode"
  timeout 600 python bench.py --no-cpu-baseline --no-extra $mode 2>gpurun_out/bench_err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],2), 'loss', d['loss']['loss'])" || tail -5 gpurun_out/bench_err.log
done 2>&1 | tee gpurun_out/r5_c21_graph.log
timeout 300 python tools/section_times.py 6 2>&1 | tail -12 | tee -a gpurun_out/r5_c21_graph.log
