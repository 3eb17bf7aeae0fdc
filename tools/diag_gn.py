import torch
import torch.nn.functional as F
dev = torch.device('cuda:0')
torch.manual_seed(0)
cl = torch.channels_last
for shape in [(2, 32, 48, 64), (2, 256, 25, 42)]:
    x = torch.randn(*shape); w = torch.randn(shape[1]); b = torch.randn(shape[1]); dy = torch.randn(*shape)
    def run(xf, dyf, dev_):
        xx = x.to(dev_).contiguous(memory_format=xf).requires_grad_()
        ww, bb = w.to(dev_).requires_grad_(), b.to(dev_).requires_grad_()
        y = F.relu(F.group_norm(xx, 8, ww, bb, 1e-5))
        g = torch.autograd.grad(y, [xx, ww, bb], dy.to(dev_).contiguous(memory_format=dyf))
        return y.detach().cpu(), [t.cpu() for t in g]
    yr, gr = run(torch.contiguous_format, torch.contiguous_format, 'cpu')
    for xf in (torch.contiguous_format, cl):
        for dyf in (torch.contiguous_format, cl):
            y, g = run(xf, dyf, dev)
            errs = [(a - r).abs().max().item() / r.abs().max().item() for a, r in zip([y] + g, [yr] + gr)]
            print(shape, 'x', 'CL' if xf == cl else 'NCHW', 'dy', 'CL' if dyf == cl else 'NCHW', ['%.1e' % e for e in errs])
