"""Losses of 16 consecutive benchmark steps (the benchmark's model, batch and step function): a fingerprint of the training
trajectory to compare two trees with (run it from each tree's root)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
sys.argv = sys.argv[:1]
import bench
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.parallel import DataParallelModel

dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
step, runner = bench.build_step(model, cfg)
data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev)
for i in range(16):
    r = step(data)
    torch.cuda.synchronize()
    lv = r['log_vars']
    print(i, ' '.join(f'{k} {float(v):.6f}' for k, v in lv.items()), flush=True)
