#!/bin/bash
# sweeps on the replayed deformable launches: long-list threshold of the per-anchor pass, pixel splits of the weight gradient
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
run() {
rm -rf /tmp/kt36; env "$@" STEP_SHAPES_REPS=4 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt36 -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/step_shapes.py > /tmp/kt36.log 2>&1 || tail -3 /tmp/kt36.log
python - <<'PY'
import csv, glob
for fn in glob.glob('/tmp/kt36/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(fn)))
    sel = [r for r in rows if any(k in r['Name'] for k in ('anchor_sum', 'anchor_combine', 'wgrad_mm', 'wgrad_reduce', 'gout_frag'))]
    print('   ' + '; '.join(f"{r['Name'].split('lsn::')[-1][:26]} {r['Calls']}x {float(r['TotalDurationNs'])/1e6:.2f} ms" for r in sel))
PY
}
for t in 40 100 20 400; do echo "== LSNET_ANCHOR_LONG=$t"; run LSNET_ANCHOR_LONG=$t; done
for s in 7 14 28; do echo "== LSNET_DCN_WGRAD_SPLITS=$s"; run LSNET_DCN_WGRAD_SPLITS=$s; done
