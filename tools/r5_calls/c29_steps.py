"""round 5: which step of the second model of a process is slow, and is it the garbage collector?  Per-step wall time (a
synchronisation after every step) of an R-101-DCN model created after an R-50 one has trained, with gc's own log."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

sys.argv = sys.argv[:1]
import bench  # noqa: E402
from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402

dev = torch.device('cuda:0')
t_gc = [0.0]


def on_gc(phase, info):
    if phase == 'start':
        t_gc[0] = time.perf_counter()
    else:
        dt = (time.perf_counter() - t_gc[0]) * 1e3
        if dt > 2:
            print(f'      gc generation {info["generation"]}: {dt:.1f} ms, collected {info["collected"]}', flush=True)


gc.callbacks.append(on_gc)


def make(backbone):
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', backbone)
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    step, _ = bench.build_step(model, cfg)
    return model, step


data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
m50, s50 = make('r50')
bench.timed_steps(s50, data, 6, 3)
m101, s101 = make('r101-dcn')
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s101(data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'step {i:2d}: enqueue {(t1 - t0) * 1e3:6.1f} ms, done after {(t2 - t0) * 1e3:6.1f} ms', flush=True)
