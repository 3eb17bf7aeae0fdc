#!/bin/bash
# experiment: weight-gradient loads interleaved with the MFMAs (alternate library built with -DLSN_WG_INTERLEAVE)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 LSNET_HIP_SO=$PWD/lsnet_amd/csrc/liblsnet_hip_il.so
timeout 60 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c26_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c26_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print('interleaved:', round(d['value'],2), round(d['ms_per_step'],2), {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
PY
timeout 45 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "(test_conv2d_matches_torch or test_dcn_forward_backward) and (bf16x6 or default)" 2>&1 | tail -2
