"""Own split-bf16 conv kernels vs MIOpen on the ResNet-50 / FPN / head shapes: accuracy and time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lsnet_amd import _lib
from lsnet_amd.ops.conv import conv2d
from tools.bench_convs import SH, timeit, B

dev = torch.device('cuda:0')
_lib.set_math_mode('bf16x3')
print(f'{"shape":36s} {"fwd own":>8s} {"MIOpen":>8s} {"TF own":>7s} {"err":>8s} | {"bwd-data own":>12s} {"MIOpen":>8s} {"err":>8s}')
tot = [0.0, 0.0, 0.0, 0.0]
for name, ci, co, k, s, h, w, cnt in SH:
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    ref = F.conv2d(x, wt, None, s, pad)
    got = conv2d(x, wt, None, s, pad)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * ref.numel() * ci * k * k
    t_own = timeit(lambda: conv2d(x, wt, None, s, pad))
    t_ref = timeit(lambda: F.conv2d(x, wt, None, s, pad))
    line = f'{name:36s} {t_own * 1e3:8.3f} {t_ref * 1e3:8.3f} {fl / t_own / 1e12:7.1f} {err:8.1e} |'
    tot[0] += t_own * cnt
    tot[1] += t_ref * cnt
    if s == 1 and co % 4 == 0:
        go = torch.randn_like(ref)
        xr = x.clone().requires_grad_()
        gref, = torch.autograd.grad(F.conv2d(xr, wt, None, s, pad), xr, go)
        xo = x.clone().requires_grad_()
        ggot, = torch.autograd.grad(conv2d(xo, wt, None, s, pad), xo, go)
        e2 = ((ggot - gref).abs().max() / gref.abs().max()).item()
        f1 = lambda: torch.ops.aten.convolution_backward(go, x, wt, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])
        ws = torch.empty(wt.numel(), device=dev)
        gx = torch.empty_like(x)
        import ctypes
        lib = _lib.load()
        cp = lambda t: ctypes.c_void_p(t.data_ptr())
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f2 = lambda: lib.lsn_conv2d_backward_data(cp(go), cp(wt), cp(gx), cp(ws), B, h, w, ci, co, k, k, s, pad, 1, st)
        t1, t2 = timeit(f1), timeit(f2)
        line += f' {t2 * 1e3:12.3f} {t1 * 1e3:8.3f} {e2:8.1e}'
        tot[2] += t2 * cnt
        tot[3] += t1 * cnt
    print(line)
print(f'network sums: forward own {tot[0] * 1e3:.1f} ms vs MIOpen {tot[1] * 1e3:.1f} ms; backward-data (stride-1 layers) own {tot[2] * 1e3:.1f} ms vs MIOpen {tot[3] * 1e3:.1f} ms')
