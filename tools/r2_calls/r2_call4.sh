#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c4_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/c4_pytest.log | tail -20
for cfg in "bf16x6 1" "bf16x3 1"; do
    set -- $cfg
    echo "== LSNET_MATH=$1 LSNET_BWD_GATHER=$2"
    LSNET_MATH=$1 LSNET_BWD_GATHER=$2 timeout 300 python tools/bench_ops.py --what dcn_all5 --iters 10 2>&1 | grep -v "^{" | tee -a gpurun_out/c4_bench_ops.log
done
( time timeout 900 python bench.py ) > gpurun_out/c4_bench.log 2>&1
echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c4_bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
    print({k:(v.get('value'), v.get('ms_per_step'), v.get('ms_per_batch')) for k,v in d.get('extra',{}).items()})
    print(d.get('cpu_baseline'))
print(open('gpurun_out/c4_bench.log').read()[-400:])
PY
