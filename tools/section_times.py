"""GPU time of the sections of one training iteration (HIP events on the launch stream, no tracer): backbone + neck forward, head
forward, loss (targets, assigners, losses), backward, clip + SGD -- and the number of kernel launches PyTorch's profiler is
NOT needed for: where the ~1 100 dispatches of a step sit relative to the time they take.
    python tools/section_times.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402
from lsnet_amd.runner import build_optimizer  # noqa: E402
from lsnet_amd.runner.hooks import OptimizerHook  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
opt = build_optimizer(model, cfg.optimizer)
hook = OptimizerHook(**dict(cfg.optimizer_config))
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
det = model.module
head = det.bbox_head


class R:   # what OptimizerHook needs of a runner
    pass


r = R()
r.model, r.optimizer, r.outputs = model, opt, {}
r.log_buffer_update = lambda *a, **k: None
names = ['backbone+neck fwd', 'head fwd', 'loss fwd', 'backward', 'clip+SGD']
acc = [0.0] * 5
for it in range(steps + 2):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    x = det.extract_feat(data['img'])
    ev[1].record()
    outs = head(x)
    ev[2].record()
    losses = head.loss(*outs, data['gt_bboxes'], data.get('gt_extremes'), None, None, data['gt_labels'], data['img_metas'])
    loss, log_vars = det._parse_losses(losses)
    ev[3].record()
    model.zero_grad_buckets()
    loss.backward()
    model.reduce_gradients()
    ev[4].record()
    r.outputs = dict(loss=loss, log_vars=log_vars, num_samples=2, backward_done=True)
    hook.after_train_iter(r)
    ev[5].record()
    torch.cuda.synchronize()
    if it >= 2:
        for i in range(5):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
print('section times per step (ms):', {n: round(a / steps, 3) for n, a in zip(names, acc)}, 'sum', round(sum(acc) / steps, 3))
