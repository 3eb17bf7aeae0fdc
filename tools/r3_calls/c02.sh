#!/bin/bash
# early-fragment-read variant: probe shapes + per-layer table + SQ counters on the 1-round shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 120 python tools/conv_probe.py > gpurun_out/c02_probe.log 2>&1
cat gpurun_out/c02_probe.log
timeout 200 python tools/bench_convs_r2.py --own-only 2>&1 | grep -v amdgpu.ids > gpurun_out/c02_convs.log
tail -32 gpurun_out/c02_convs.log | cut -c1-150
cd /tmp && export TMPDIR=/tmp
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA TCP_TCC_READ_REQ_sum TCC_HIT_sum"; do
  rm -rf /tmp/pmc_c02
  CONV_REPS=3 timeout 120 rocprofv3 --pmc $PMC --kernel-trace -d /tmp/pmc_c02 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_probe.py 1round:256:256:3:1:128:128 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_c02/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'conv_mm_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'  {k:28s} n={len(v)} mean {sum(v) / len(v):.4g}')
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/c02_pmc.log
