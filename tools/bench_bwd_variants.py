"""Where does dcn_backward spend its time?  Times the C-ABI backward with different gradient subsets."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd.ops import get_backend

dev = torch.device('cuda:0')
cl = torch.channels_last
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C = 2, 256
torch.manual_seed(0)
be = get_backend(torch.zeros(1, device=dev))
w = (torch.randn(C, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=cl)
xs = [torch.randn(B, C, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
offs = [(torch.randn(B, 18, h, ww, device=dev) * scale).contiguous(memory_format=cl) for h, ww in LEVELS]
msks = [torch.rand(B, 9, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
gos = [torch.randn(B, C, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
cfg = dict(stride=1, pad=1, dil=1, groups=1, dg=1, scales=[(1.0, 1.0)] * 5, pyramid=False)
fl = sum(2.0 * B * h * ww * C * C * 9 for h, ww in LEVELS)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def need(inp, off, wgt):
    return dict(input=[inp] * 5, offset=[off] * 5, mask=[off] * 5, weight=wgt, bias=wgt)


for name, nd in (('all', need(True, True, True)), ('input only', need(True, False, False)),
                 ('offset+mask only', need(False, True, False)), ('input+offset', need(True, True, False)),
                 ('weight only', need(False, False, True))):
    t = timeit(lambda: be.dcn_backward(xs, offs, msks, w, gos, cfg, nd))
    print(f'{name:20s} {t:8.3f} ms   (data GEMM {fl / 1e9:.1f} GF -> {fl / t / 1e9:.1f} TF/s if that were all)')
t = timeit(lambda: be.dcn_forward(xs, offs, msks, w, None, cfg, LEVELS))
print(f'{"forward":20s} {t:8.3f} ms   {fl / t / 1e9:.1f} TF/s')
