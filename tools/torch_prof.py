"""Which ATen ops make up the small-kernel soup of a training step (torch.profiler, one step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
sys.argv = sys.argv[:1]
import bench

dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = model.to(dev).to(memory_format=torch.channels_last).train()
step, runner = bench.build_step(model, cfg)
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev)
for _ in range(3):
    step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(data)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=48))
