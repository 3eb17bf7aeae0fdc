"""round 5: the R-101-DCN step is 50 ms in a process of its own and 67 ms as the last extra leg of bench.py -- after which of the
earlier legs?  The legs of bench.py one after the other, the fixed-shape R-101-DCN step timed (fresh model) in between."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

sys.argv = sys.argv[:1]
import bench  # noqa: E402
from lsnet_amd import _lib  # noqa: E402
from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402

dev = torch.device('cuda:0')


def r101(tag):
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r101-dcn')
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    step, _ = bench.build_step(model, cfg)
    data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
    dt = bench.timed_steps(step, data, 5, 4)
    st = _lib.scratch_stats()
    print(f'{tag:34s} R-101-DCN {dt * 1e3:6.2f} ms/step   library scratch {st["held_bytes"] / 2 ** 20:7.0f} MB, '
          f'torch reserved {torch.cuda.memory_reserved() / 2 ** 30:5.1f} GB', flush=True)
    del model, step, data
    torch.cuda.empty_cache()


r101('fresh process')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
step, runner = bench.build_step(model, cfg)
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
bench.timed_steps(step, data, 6, 3)
r101('after the R-50 steps')
timer = bench.KernelTimer()
timer.start()
bench.timed_steps(step, data, 3, 0)
timer.stop()
r101('after a kernel-timer survey')
for mode in ('bf16x3', 'fp32'):
    _lib.set_math_mode(mode)
    bench.timed_steps(step, data, 3, 2)
_lib.set_math_mode('bf16x6')
r101('after the other math modes')
bench.infer_leg(dev)
r101('after the inference leg')
bench.config_leg(dev, 'segm', 'x101-dcn', n=2, warm=2)
r101('after the X-101 leg')
