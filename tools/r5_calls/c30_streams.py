"""round 5: does the speed of the step depend on WHICH stream the head's second stream is?  Ten models one after the other in one
process, the R-50 step timed back to back.  (With one stream per LSHead instance: profiles/r5_stream_queues.txt, 50 - 54 ms for the 2nd,
3rd, 7th model; with ops/streams.py's one stream per process every model runs alike.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

sys.argv = sys.argv[:1]
import bench  # noqa: E402
from lsnet_amd.data import synthetic_batch  # noqa: E402
from lsnet_amd.model_zoo import build_lsnet  # noqa: E402
from lsnet_amd.parallel import DataParallelModel  # noqa: E402

dev = torch.device('cuda:0')
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
keep = []
for k in range(10):
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r50')
    model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    step, _ = bench.build_step(model, cfg)
    dt = bench.timed_steps(step, data, 6, 4)
    from lsnet_amd.ops import streams
    print(f'model {k}: {dt * 1e3:6.2f} ms/step; second stream(s) of the process {[hex(s.cuda_stream) for s in streams._side.values()]}',
          flush=True)
    del model, step
    torch.cuda.empty_cache()
