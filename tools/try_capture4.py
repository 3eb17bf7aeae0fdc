"""Debug helper: GraphedForwardBackward + selectable eager work between replays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.runner.graph_step import GraphedForwardBackward

what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = model.to(dev).to(memory_format=torch.channels_last).train()
data = synthetic_batch('bbox', 2, 800, 1344, seed=40, device=dev)
gs = GraphedForwardBackward(model, warmup=2)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=0.0, momentum=0.9, weight_decay=1e-4)
acc = None
bufs = [torch.zeros_like(p) for p in params]
g2 = None
def foreach_step():
    grads = [p.grad for p in params]
    d = torch._foreach_add(grads, params, alpha=1e-4)
    torch._foreach_mul_(bufs, 0.9)
    torch._foreach_add_(bufs, d)
    torch._foreach_add_(params, bufs, alpha=-0.0)
for i in range(n):
    out = gs(data)
    if 'pstep' in what:
        for p, b in zip(params, bufs):
            b.mul_(0.9).add_(p.grad).add_(p, alpha=1e-4)
            p.data.add_(b, alpha=-0.0)
    if 'fstep' in what:
        foreach_step()
    if 'gstep' in what and gs.graph is not None:
        if g2 is None:
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                foreach_step()
        g2.replay()
    if 'clip' in what:
        torch.nn.utils.clip_grad_norm_(params, 35.0)
    if what == 'step':
        opt.step()
    if 'log' in what:
        v = out['log_vars']['loss'] * 2
        acc = v if acc is None else acc + v
    if 'norm' in what:
        t = torch.stack([p.grad.norm() for p in params]).norm()
    torch.cuda.synchronize()
    if 'chk' in what:
        pf = [n for (n, p) in model.named_parameters() if p.requires_grad and not bool(p.isfinite().all())]
        gf = [n for (n, p) in model.named_parameters() if p.requires_grad and not bool(p.grad.isfinite().all())]
        gmax = max(float(p.grad.abs().max()) for p in params)
        bmax = max(float(b.abs().max()) for b in bufs)
        print('   nonfinite params', pf[:5], 'grads', gf[:5], 'gmax %.3g bmax %.3g' % (gmax, bmax), flush=True)
    print(what, i, float(out['loss']), flush=True)
print(what, 'OK')
