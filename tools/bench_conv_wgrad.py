"""Own conv weight-gradient kernel (deformable-conv wgrad kernel without offsets) vs MIOpen."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lsnet_amd import _lib
from tools.bench_convs import SH, timeit, B

dev = torch.device('cuda:0')
lib = _lib.load()
cp = lambda t: ctypes.c_void_p(t.data_ptr())
tot = [0.0, 0.0]
for name, ci, co, k, s, h, w, cnt in SH:
    if ci < 64 or name.startswith('l1'):
        continue
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    y = F.conv2d(x, wt, None, s, pad)
    go = torch.randn_like(y)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gw = torch.empty_like(wt)
    gb = torch.empty(co, device=dev)
    f_own = lambda: lib.lsn_conv2d_backward_weight(cp(x), cp(go), cp(gw), cp(gb), B, h, w, ci, co, k, k, s, pad, 1, 0, st)
    f_ref = lambda: torch.ops.aten.convolution_backward(go, x, wt, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])
    assert f_own() == 0, lib.lsn_last_error()
    ref = f_ref()[1]
    err = ((gw - ref).abs().max() / ref.abs().max()).item()
    eb = ((gb - go.sum((0, 2, 3))).abs().max() / go.sum((0, 2, 3)).abs().max()).item()
    t1, t2 = timeit(f_own), timeit(f_ref)
    fl = 2.0 * y.numel() * ci * k * k
    tot[0] += t1 * cnt; tot[1] += t2 * cnt
    print(f'{name:36s} own {t1 * 1e3:7.3f} ms ({fl / t1 / 1e12:6.1f} TF)  MIOpen {t2 * 1e3:7.3f} ms ({fl / t2 / 1e12:6.1f} TF)  err {err:.1e} bias {eb:.1e}')
print(f'sum: own {tot[0] * 1e3:.1f} ms, MIOpen {tot[1] * 1e3:.1f} ms')
