"""Debug helper: which part of the step breaks under repeated hipGraph replay."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = model.to(dev).to(memory_format=torch.channels_last).train()
data = synthetic_batch('bbox', 2, 800, 1344, seed=40, device=dev)
params = [p for p in model.parameters() if p.requires_grad]


def fn():
    if mode == 'fwd':
        with torch.no_grad():
            return model.train_step(data, None)['loss']
    if mode == 'backbone':
        feats = model.extract_feat(data['img'])
        loss = sum(f.square().mean() for f in feats)
        loss.backward()
        return loss
    if mode == 'headfwd':
        with torch.no_grad():
            feats = model.extract_feat(data['img'])
            outs = model.bbox_head(feats)
            return sum(o.mean() for o in outs[0])
    if mode == 'head':   # head forward/backward from detached features
        with torch.no_grad():
            feats = model.extract_feat(data['img'])
        outs = model.bbox_head([f.detach().requires_grad_() for f in feats])
        loss = sum(o.square().mean() for o in outs[0]) + sum(o.square().mean() for o in outs[2])
        loss.backward()
        return loss
    out = model.train_step(data, None)
    out['loss'].backward()
    return out['loss']


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        for p in params:
            p.grad = None
        fn()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
for p in params:
    p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    loss = fn()
for i in range(n):
    g.replay()
    torch.cuda.synchronize()
    print(mode, i, float(loss), 'alloc %.2f GB' % (torch.cuda.memory_allocated() / 2 ** 30), flush=True)
print(mode, 'OK')
