"""round 5: DataParallelModel creates the step's second stream and warms the library's own before anything else: three models in a
row (the step must run as before), and the warm-up must not raise."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

sys.argv = sys.argv[:1]
import bench  # noqa: E402
from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.ops import streams  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402

dev = torch.device('cuda:0')
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    for k in range(3):
        torch.manual_seed(0)
        model, cfg = build_lsnet('bbox', 'r50')
        model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
        step, _ = bench.build_step(model, cfg)
        dt = bench.timed_steps(step, data, 6, 4)
        out = step(data)
        print(f'model {k}: {dt * 1e3:6.2f} ms/step, loss {float(out["log_vars"]["loss"]):.5f}, warmed {sorted(streams._warm)}', flush=True)
        del model, step
        torch.cuda.empty_cache()
print('warnings about the warm-up:', [str(x.message)[:200] for x in w if 'warm-up' in str(x.message)])
