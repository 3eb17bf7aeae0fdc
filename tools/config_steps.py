"""A few training steps of another BASELINE config (default: config 4 = X-101-64x4d-DCN segm) between the two marker dispatches
tools/prof_summary.py looks for: the workload of the rocprofv3 kernel traces profiles/r6_cfg{3,4}_kernel_stats.txt.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cfg4 -o t -- python tools/config_steps.py segm x101-dcn 3
    python tools/prof_summary.py /tmp/prof_cfg4 profiles/r6_cfg4_kernel_stats.txt 3
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch  # noqa: E402

task, backbone, steps = sys.argv[1] if len(sys.argv) > 1 else 'segm', sys.argv[2] if len(sys.argv) > 2 else 'x101-dcn', \
    int(sys.argv[3]) if len(sys.argv) > 3 else 3
sys.argv = sys.argv[:1]
import bench  # noqa: E402
from lsnet_amd import _lib  # noqa: E402
from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.ops import get_backend  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
if os.environ.get('CONFIG_STEPS_DBG'):   # debug word of the library for the whole run (A/B arms, tools/README.md)
    _lib.load().lsn_debug_phase_clocks(None, int(os.environ['CONFIG_STEPS_DBG']))
model, cfg = build_lsnet(task, backbone)
model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
step, _ = bench.build_step(model, cfg)
data = synthetic_batch(task, 2, 800, 1344, seed=1234, device=dev, channels_last=True)
dt = bench.timed_steps(step, data, steps, 3)
print(f'{task} {backbone}: {dt * 1e3:.2f} ms/step untraced-in-process ({2 / dt:.2f} img/s)', flush=True)
# per-family event log of the library (bench.py's survey), then the marked steps
timer = bench.KernelTimer()
timer.start()
for _ in range(steps):
    step(data)
ks = timer.stop()
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]['total_ms']):
    print(f"  {k:14s} {v['total_ms'] / steps:8.3f} ms/step  {v['launches'] // steps:4d} launches/step  {v['tflops']:7.1f} TF  "
          f"{v['alg_gbps']:8.1f} GB/s alg", flush=True)
mk = torch.ones(32, 32, device=dev)
get_backend(mk).selftest_mfma(mk, mk, 0)
torch.cuda.synchronize()
for _ in range(steps):
    step(data)
torch.cuda.synchronize()
get_backend(mk).selftest_mfma(mk, mk, 0)
torch.cuda.synchronize()
