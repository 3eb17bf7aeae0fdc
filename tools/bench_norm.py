"""Bandwidth of the fused norm kernels on the backbone / head tensor sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd.ops.batch_norm import bn_act
from lsnet_amd.ops.group_norm import GroupNorm
dev = torch.device('cuda:0')


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for C, h, w in ((256, 200, 336), (64, 200, 336), (512, 100, 168), (1024, 50, 84)):
    bn = torch.nn.BatchNorm2d(C).to(dev).eval()
    x = torch.randn(2, C, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    r = torch.randn_like(x).requires_grad_()
    y = bn_act(bn, x, relu=True, residual=r)
    go = torch.randn_like(y)
    nb = x.numel() * 4
    tf = timeit(lambda: bn_act(bn, x, relu=True, residual=r))
    tb = timeit(lambda: torch.autograd.grad(bn_act(bn, x, relu=True, residual=r), [x, r, bn.weight, bn.bias], go)) - tf
    print(f'bn+add+relu C={C:5d} {h}x{w}: fwd {tf * 1e6:7.1f} us ({3 * nb / tf / 1e12:.2f} TB/s)  bwd {tb * 1e6:7.1f} us ({5 * nb / tb / 1e12:.2f} TB/s)')
gn = GroupNorm(32, 256).to(dev)
xs = [torch.randn(2, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_() for h, w in ((100, 168), (50, 84), (25, 42), (13, 21), (7, 11))]
gos = [torch.randn_like(x) for x in xs]
nb = sum(x.numel() for x in xs) * 4
tf = timeit(lambda: gn.forward_multi(xs, relu=True))
tb = timeit(lambda: torch.autograd.grad(gn.forward_multi(xs, relu=True), xs + [gn.weight, gn.bias], gos)) - tf
print(f'gn+relu 5 levels: fwd {tf * 1e6:7.1f} us ({3 * nb / tf / 1e12:.2f} TB/s)  bwd {tb * 1e6:7.1f} us ({5 * nb / tb / 1e12:.2f} TB/s)')
