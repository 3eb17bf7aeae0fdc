"""Debug helper: warm up, capture forward+backward of LSNet R-50 into a hipGraph, replay, compare with eager."""
import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.runner.graph_step import GraphedForwardBackward

dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = model.to(dev).to(memory_format=torch.channels_last).train()
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 480)
data = synthetic_batch('bbox', 2, h, w, seed=40, device=dev)
gs = GraphedForwardBackward(model, warmup=2)
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 5):
    out = gs(data)
    torch.cuda.synchronize()
    print(i, 'graph' if gs.graph is not None else 'eager', float(out['loss']), flush=True)
