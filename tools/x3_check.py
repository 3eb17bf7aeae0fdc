"""Accuracy and speed of the split-bf16 forward kernel against the exact fp32 MFMA kernel (same inputs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd import _lib
from lsnet_amd.ops import get_backend

dev = torch.device('cuda:0')
cl = torch.channels_last
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C = 2, 256
torch.manual_seed(0)
be = get_backend(torch.zeros(1, device=dev))
w = (torch.randn(C, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=cl)
bias = torch.randn(C, device=dev) * 0.1
xs = [torch.randn(B, C, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
offs = [(torch.randn(B, 18, h, ww, device=dev) * 0.5).contiguous(memory_format=cl) for h, ww in LEVELS]
msks = [torch.rand(B, 9, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
cfg = dict(stride=1, pad=1, dil=1, groups=1, dg=1, scales=[(1.0, 1.0)] * 5, pyramid=False)


def run():
    return be.dcn_forward(xs, offs, msks, w, bias, cfg, LEVELS)


def timeit(n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


_lib.set_math_mode('fp32')
ref = [o.clone() for o in run()]
t_ref = timeit()
_lib.set_math_mode('bf16x3')
got = [o.clone() for o in run()]
t_x3 = timeit()
fl = sum(2.0 * B * h * ww * C * C * 9 for h, ww in LEVELS)
for (h, ww), r, g in zip(LEVELS, ref, got):
    d = (r - g).abs()
    print(f'level {h}x{ww}: max abs err {d.max().item():.3e}  / max |ref| {r.abs().max().item():.3f} = {d.max().item() / r.abs().max().item():.2e};'
          f' rms err / rms ref = {(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item():.2e}')
print(f'fp32 MFMA  : {t_ref:.3f} ms  {fl / t_ref / 1e9:.1f} TF')
print(f'split bf16 : {t_x3:.3f} ms  {fl / t_x3 / 1e9:.1f} TF (algorithmic)')
