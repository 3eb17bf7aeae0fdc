"""Where the small ATen kernels of a training step come from: (op, first lsnet_amd / bench frame) -> launches, device time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import defaultdict
from torch.profiler import profile, ProfilerActivity
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
sys.argv = sys.argv[:1]
import bench

dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = model.to(dev).to(memory_format=torch.channels_last).train()
from lsnet_amd.parallel import DataParallelModel
model = DataParallelModel(model)
step, runner = bench.build_step(model, cfg)
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev)
for _ in range(3):
    step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(data)
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
shapes = defaultdict(lambda: [0, 0.0])
detail = defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(e, 'self_cuda_time_total', 0)
    if not dt or not e.name.startswith('aten::'):
        continue
    site = 'autograd engine / other'
    frames = [fr for fr in (e.stack or []) if ('lsnet_amd' in fr or 'bench.py' in fr) and 'torch/' not in fr]
    if frames:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/'
        site = ' <- '.join(f.replace(root, '').replace('lsnet_amd/', '') for f in frames[:2])
    if 'hooks.py' in site or site.startswith('autograd'):
        sh = shapes[(e.name, str(getattr(e, 'input_shapes', None))[:90])]
        sh[0] += 1
        sh[1] += dt
    a = agg[(e.name, site)]
    a[0] += 1
    a[1] += dt
    if e.name in ('aten::copy_', 'aten::fill_', 'aten::cat', 'aten::mul', 'aten::add', 'aten::index'):
        d = detail[(e.name, site[:60], str(getattr(e, 'input_shapes', None))[:100])]
        d[0] += 1
        d[1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot_n = sum(v[0] for v in agg.values()); tot_t = sum(v[1] for v in agg.values())
print(f'ATen ops with device time in one step: {tot_n} launches, {tot_t / 1e3:.2f} ms')
for (name, site), (n, t) in rows[:70]:
    print(f'{n:5d} {t / 1e3:8.3f} ms  {name:28s} {site[:110]}')
print('--- ops launched from backward() / the autograd engine, by input shapes')
for (name, shp), (n, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{n:5d} {t / 1e3:8.3f} ms  {name:24s} {shp}')
print('--- copies / fills / cats / muls by site and input shapes')
for (name, site, shp), (n, t) in sorted(detail.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{n:5d} {t / 1e3:8.3f} ms  {name:14s} {site:60s} {shp}')
