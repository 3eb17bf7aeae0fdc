import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd import _lib
from lsnet_amd.ops import get_backend
dev = torch.device('cuda:0'); cl = torch.channels_last
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C = 2, 256
be = get_backend(torch.zeros(1, device=dev)); lib = _lib.load()
w = (torch.randn(C, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=cl)
xs = [torch.randn(B, C, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
offs = [(torch.randn(B, 18, h, ww, device=dev) * 0.5).contiguous(memory_format=cl) for h, ww in LEVELS]
msks = [torch.rand(B, 9, h, ww, device=dev).contiguous(memory_format=cl) for h, ww in LEVELS]
cfg = dict(stride=1, pad=1, dil=1, groups=1, dg=1, scales=[(1.0, 1.0)] * 5, pyramid=False)
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, fl in (('full', 0), ('no W loads', 1 << 20), ('no x loads', 1 << 21), ('no loads at all', 3 << 20)):
    lib.lsn_debug_phase_clocks(None, fl)
    print(f'{name:18s} {timeit(lambda: be.dcn_forward(xs, offs, msks, w, None, cfg, LEVELS)):.3f} ms')
lib.lsn_debug_phase_clocks(None, 0)
