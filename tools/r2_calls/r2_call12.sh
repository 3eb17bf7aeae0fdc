#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_variants_gpu.py tests/test_golden_gpu.py tests/test_zz_cpv_gpu.py -q -s --tb=short -p no:cacheprovider -k "grouped or x101 or cpv or res2net" > gpurun_out/c12_pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed|^FAILED|^ERROR|Error|worst" gpurun_out/c12_pytest.log | tail -14
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 > gpurun_out/c12_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/c12_bench.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), {k:(round(v['avg_ms'],3), round(v['tflops'],1)) for k,v in d.get('kernels',{}).items()})
PY
python - <<'PY'
# grouped conv timing at the X-101 64x4d layer1 shape vs ATen
import torch, torch.nn.functional as F
from lsnet_amd.ops.conv import Conv2d
dev='cuda:0'
for C,G,H,W,s in [(256,64,200,336,1),(512,64,200,336,2),(1024,64,100,168,2),(2048,64,50,84,2)]:
    m=Conv2d(C,C,3,stride=s,padding=1,groups=G,bias=False).to(dev).to(memory_format=torch.channels_last)
    x=torch.randn(2,C,H,W,device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    def own():
        y=m(x); y.backward(torch.ones_like(y)); m.weight.grad=None; x.grad=None
    def aten():
        y=F.conv2d(x,m.weight,None,s,1,1,G); y.backward(torch.ones_like(y)); m.weight.grad=None; x.grad=None
    for name,fn in (('own',own),('aten',aten)):
        for _ in range(2): fn()
        torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): fn()
        b.record(); torch.cuda.synchronize()
        print(f'grouped 3x3 C={C} G={G} {H}x{W} s{s}: {name} fwd+bwd {a.elapsed_time(b)/5:.3f} ms')
PY
