#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "pyramid_launch or tower_launch or deterministic or pyr_" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for thr in 400 0; do
echo "== LSNET_GATHER_ANCHOR=$thr"
rm -rf /tmp/kt19; LSNET_GATHER_ANCHOR=$thr STEP_SHAPES_REPS=4 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt19 -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/step_shapes.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for fn in glob.glob('/tmp/kt19/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(fn))):
        if 'gather' in r['Name'] or 'anchor' in r['Name']:
            print('  ', r['Name'][:64].ljust(64), r['Calls'].rjust(4), f"{float(r['AverageNs'])/1e3:9.1f} us avg")
PY
done
