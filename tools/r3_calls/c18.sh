#!/bin/bash
# gather: long-list blocks first.  Kernel trace of the step's DCN launch shapes + operator tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt18; STEP_SHAPES_REPS=4 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt18 -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/step_shapes.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for fn in glob.glob('/tmp/kt18/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(fn)))[:14]:
        print('  ', r['Name'][:64].ljust(64), r['Calls'].rjust(4), f"{float(r['AverageNs'])/1e3:9.1f} us avg", f"{float(r['MinNs'])/1e3:8.1f} min {float(r['MaxNs'])/1e3:8.1f} max")
PY
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "dcn or tower or pyramid or gather" 2>&1 | tail -2
