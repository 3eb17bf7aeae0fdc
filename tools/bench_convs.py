"""MIOpen (ATen conv2d, channels_last fp32) forward / backward rates on the ResNet-50 + FPN + head conv shapes of
the benchmark step (B=2, 800x1344): the numbers an own implicit-GEMM kernel has to beat."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
B = 2
# (name, Cin, Cout, k, stride, H, W (input), count in the network)
SH = [
    ('l1 1x1 64->64', 64, 64, 1, 1, 200, 336, 3), ('l1 3x3 64', 64, 64, 3, 1, 200, 336, 3),
    ('l1 1x1 64->256', 64, 256, 1, 1, 200, 336, 4), ('l1 1x1 256->64', 256, 64, 1, 1, 200, 336, 2),
    ('l2 1x1 256->128', 256, 128, 1, 1, 200, 336, 1), ('l2 3x3 s2 128', 128, 128, 3, 2, 200, 336, 1),
    ('l2 3x3 128', 128, 128, 3, 1, 100, 168, 3), ('l2 1x1 128->512', 128, 512, 1, 1, 100, 168, 4),
    ('l2 1x1 512->128', 512, 128, 1, 1, 100, 168, 3), ('l2 ds 1x1 s2 256->512', 256, 512, 1, 2, 200, 336, 1),
    ('l3 1x1 512->256', 512, 256, 1, 1, 100, 168, 1), ('l3 3x3 s2 256', 256, 256, 3, 2, 100, 168, 1),
    ('l3 3x3 256', 256, 256, 3, 1, 50, 84, 5), ('l3 1x1 256->1024', 256, 1024, 1, 1, 50, 84, 6),
    ('l3 1x1 1024->256', 1024, 256, 1, 1, 50, 84, 5),
    ('l4 1x1 1024->512', 1024, 512, 1, 1, 50, 84, 1), ('l4 3x3 s2 512', 512, 512, 3, 2, 50, 84, 1),
    ('l4 3x3 512', 512, 512, 3, 1, 25, 42, 2), ('l4 1x1 512->2048', 512, 2048, 1, 1, 25, 42, 3),
    ('l4 1x1 2048->512', 2048, 512, 1, 1, 25, 42, 2),
    ('fpn lat 512->256', 512, 256, 1, 1, 100, 168, 1), ('fpn 3x3 256 P3', 256, 256, 3, 1, 100, 168, 1),
    ('head 3x3 256 P3 (x8 convs)', 256, 256, 3, 1, 100, 168, 8), ('head 1x1 768->256 P3', 768, 256, 1, 1, 100, 168, 2),
    ('head 3x3 256->27 P3 (offset conv)', 256, 27, 3, 1, 100, 168, 6),
]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


tot_f = tot_b = 0.0
print(f'{"shape":38s} {"fwd ms":>8s} {"TF":>6s} {"bwd ms":>8s} {"TF":>6s}  x count')
for name, ci, co, k, s, h, w, cnt in SH:
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    wt = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_()
    pad = k // 2
    y = F.conv2d(x, wt, None, s, pad)
    go = torch.randn_like(y)
    fl = 2.0 * y.numel() * ci * k * k
    tf = timeit(lambda: F.conv2d(x, wt, None, s, pad))

    def bwd():
        torch.autograd.grad(F.conv2d(x, wt, None, s, pad), (x, wt), go)
    tb = timeit(bwd) - tf
    tot_f += tf * cnt
    tot_b += tb * cnt
    print(f'{name:38s} {tf * 1e3:8.3f} {fl / tf / 1e12:6.1f} {tb * 1e3:8.3f} {2 * fl / tb / 1e12:6.1f}  x{cnt}')
print(f'sum over the network: forward {tot_f * 1e3:.1f} ms, backward {tot_b * 1e3:.1f} ms')
