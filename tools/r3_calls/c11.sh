#!/bin/bash
# wgrad kernel anatomy on the head 3x3 shape: kernel trace (main vs reduce), SQ counters, split sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 CONV_PASS=wgrad
for S in 0 8 16 32 64; do echo "== LSNET_WGRAD_SPLITS=$S"; LSNET_WGRAD_SPLITS=$S timeout 60 python tools/conv_probe.py P3:256:256:3:1:100:168 all5:256:256:3:1:140:160 l3:256:256:3:1:50:84 1x1:768:256:1:1:100:168 2>&1 | grep -v amdgpu; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; CONV_REPS=5 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_probe.py P3:256:256:3:1:100:168 > /dev/null 2>&1
python - <<'PY'
import csv, glob
for fn in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(fn)))[:6]:
        print('  ', r['Name'][:70], r['Calls'], r['AverageNs'])
PY
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_c11
  CONV_REPS=3 timeout 120 rocprofv3 --pmc $PMC --kernel-trace -d /tmp/pmc_c11 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_probe.py P3:256:256:3:1:100:168 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob('/tmp/pmc_c11/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'conv_wgrad_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'  {k:28s} n={len(v)} mean {sum(v) / len(v):.4g}')
PY
done
