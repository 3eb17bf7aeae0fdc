#!/bin/bash
# backward-data as the dense 1x1 kernel + corner sums in the gather pass: parity, determinism, kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "(test_dcn_forward_backward and (default or x3_gather or first_gemms)) or deterministic or tower_launch or pyramid_launch or multi_level or entry_points or dcn_pack" 2>&1 | grep -v "amdgpu\|Warn\|warn" | tail -25
cd /tmp && export TMPDIR=/tmp
for mm in 1 0; do
echo "== LSNET_DCN_MM=$mm"
rm -rf /tmp/kt24; LSNET_DCN_MM=$mm STEP_SHAPES_REPS=4 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt24 -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/step_shapes.py > /tmp/kt24.log 2>&1 || tail -5 /tmp/kt24.log
python - <<'PY'
import csv, glob
for fn in glob.glob('/tmp/kt24/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(fn)))
    tot = sum(float(r['TotalDurationNs']) for r in rows if 'lsn::' in r['Name'] or 'rocprim' in r['Name'])
    print('   lsn kernels total %.2f ms' % (tot / 1e6))
    for r in rows[:12]:
        print('  ', r['Name'][:70].ljust(70), r['Calls'].rjust(4), f"{float(r['AverageNs'])/1e3:9.1f} us avg", f"{float(r['MinNs'])/1e3:8.1f} min {float(r['MaxNs'])/1e3:8.1f} max")
PY
done
