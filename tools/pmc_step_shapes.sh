#!/bin/bash
# HBM counters (FETCH_SIZE, WRITE_SIZE: separate passes, as MI355X_MICROARCH.md prescribes) of the deformable-conv launches
# of one benchmark step, replayed by tools/step_shapes.py (tower launch over the 5 FPN levels, pyramid launch over the 15
# pairs).  Hard 120 s limit per pass.  Writes gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.txt and <tag>_hbm_traffic.json.
set -u
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
    raw=/tmp/pmcstep_${tag}_$ctr
    rm -rf "$raw"
    timeout -s KILL 120 rocprofv3 --pmc $ctr --kernel-include-regex 'lsn::|rocprim' --output-format csv -d "$raw" -o ops -- \
        python tools/step_shapes.py > gpurun_out/${tag}_pmc_${ctr}_run.log 2>&1
    echo "pass $ctr exit $?"
done
python tools/pmc_traffic.py /tmp/pmcstep_${tag}_FETCH_SIZE /tmp/pmcstep_${tag}_WRITE_SIZE gpurun_out/${tag}_hbm_traffic.json \
    > gpurun_out/${tag}_pmc_hbm.txt 2>&1
cat gpurun_out/${tag}_pmc_hbm.txt
