#!/bin/bash
# HBM and SQ counters of the dense-conv kernels (conv_mm_kernel forward, conv_wgrad_kernel) at shapes of the benchmark
# step, one counter group per pass (rocprofv3 --pmc; FETCH_SIZE and WRITE_SIZE in separate passes, FETCH doubled on
# gfx950 as MI355X_MICROARCH.md prescribes).  Writes gpurun_out/<tag>_pmc_conv.txt.
set -u
tag=${1:-r3}
cd /tmp && export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
out=$ROOT/gpurun_out/${tag}_pmc_conv.txt
mkdir -p $ROOT/gpurun_out
: > $out
SHAPES="head3x3_all5:256:256:3:1:140:160 l2_3x3:128:128:3:1:100:168 l1_1x1:64:256:1:1:200:336 l3_1x1:1024:256:1:1:50:84"
for pass in fwd wgrad; do
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pmc_conv
    CONV_PASS=$pass CONV_REPS=2 timeout -s KILL 120 rocprofv3 --pmc $PMC --kernel-include-regex 'conv_mm_kernel|conv_wgrad_kernel' \
        --output-format csv -d /tmp/pmc_conv -o p -- python $ROOT/tools/conv_probe.py $SHAPES > /dev/null 2>&1
    PASS=$pass python - >> $out <<'PY'
import csv, glob, collections, os
acc = collections.OrderedDict()
for fn in glob.glob('/tmp/pmc_conv/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        key = (r['Kernel_Name'].split('(')[0][-60:], r['Grid_Size'], r['Counter_Name'])
        acc.setdefault(key, []).append(float(r['Counter_Value']))
for (k, g, c), v in acc.items():
    print(f"{os.environ['PASS']:6s} {k:62s} grid {g:>8s} {c:28s} n={len(v)} mean {sum(v) / len(v):.5g}")
PY
  done
done
cat $out
