#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -p no:cacheprovider -s \
    -k "bench_shape or split6 or deterministic or conv_split6" > gpurun_out/c2_pytest_a.log 2>&1
echo "pytest A rc $?"; grep -E "passed|failed" gpurun_out/c2_pytest_a.log | tail -3
timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=line -p no:cacheprovider -x \
    -k "not bench_shape and not split6 and not deterministic and not conv_split6" > gpurun_out/c2_pytest_b.log 2>&1
echo "pytest B rc $?"; tail -n 5 gpurun_out/c2_pytest_b.log
for cfg in "bf16x6 1" "bf16x3 1"; do
    set -- $cfg
    echo "== LSNET_MATH=$1 LSNET_BWD_GATHER=$2"
    LSNET_MATH=$1 LSNET_BWD_GATHER=$2 timeout 300 python tools/bench_ops.py --what dcn_all5 --iters 10 2>&1 | grep -v "^{" | tee -a gpurun_out/c2_bench_ops.log
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/c2_bench.log 2>&1
echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c2_bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
else:
    print(open('gpurun_out/c2_bench.log').read()[-2000:])
PY
