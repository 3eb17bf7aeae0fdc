"""Is the eager training step bound by the CPU's launch rate anywhere?  (1) every synchronising call of one step
(torch.cuda.set_sync_debug_mode); (2) the CPU time to ENQUEUE a step against the GPU time to run it: steps issued back to back
without a synchronisation in between -- if the CPU needs less, it runs ahead and the GPU never waits for a launch.
    python tools/cpu_lead.py [steps]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sys.argv, steps = sys.argv[:1], int(sys.argv[1]) if len(sys.argv) > 1 else 10
import bench  # noqa: E402
from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
step, runner = bench.build_step(model, cfg)
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
for _ in range(4):
    step(data)
torch.cuda.synchronize()

torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step(data)
    torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('default')
sync = [x for x in w if 'synchroniz' in str(x.message).lower()]
print(f'synchronising calls in one step: {len(sync)}')
for x in sync[:20]:
    print('   ', x.filename.split('/root/repo/')[-1], x.lineno, str(x.message)[:120])

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
t0 = time.perf_counter()
marks = []
for _ in range(steps):
    step(data)
    marks.append(time.perf_counter())
t_cpu = time.perf_counter() - t0
e1.record()
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'{steps} steps: CPU enqueue {t_cpu / steps * 1e3:.2f} ms/step, GPU {e0.elapsed_time(e1) / steps:.2f} ms/step, '
      f'wall {t_all / steps * 1e3:.2f} ms/step')
print('CPU time of each step (ms):', ' '.join(f'{(b - a) * 1e3:.1f}' for a, b in zip([t0] + marks[:-1], marks)))
