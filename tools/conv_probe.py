"""Dense conv kernel probe: time / TFLOP/s of chosen shapes, for tile-quantisation and steady-state checks.
usage: conv_probe.py [name:C:Co:k:stride:H:W ...]   (B = 2);  env CONV_REPS, CONV_PASS=fwd|wgrad"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from lsnet_amd import _lib
from lsnet_amd.ops.conv import conv2d

dev = torch.device('cuda:0')
B = 2
DEFAULT = ['1round:256:256:3:1:128:128', '2rounds:256:256:3:1:128:256', 'P3:256:256:3:1:100:168', 'all5:256:256:3:1:140:160',
           'l2_3x3:128:128:3:1:100:168', 'l3_3x3:256:256:3:1:50:84', 'l4_3x3:512:512:3:1:25:42', 'l1_1x1:64:256:1:1:200:336',
           'l4_1x1:2048:512:1:1:25:42']
reps = int(os.environ.get('CONV_REPS', 20))
for spec in (sys.argv[1:] or DEFAULT):
    name, C, Co, k, s, H, W = spec.split(':')
    C, Co, k, s, H, W = map(int, (C, Co, k, s, H, W))
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    if os.environ.get('CONV_PASS') == 'wgrad':
        lib = _lib.load()
        cp = lambda t: ctypes.c_void_p(t.data_ptr())
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        with torch.no_grad():
            y = conv2d(x, w, None, s, k // 2)
        go, gw = torch.randn_like(y), torch.empty_like(w)
        f = lambda: lib.lsn_conv2d_backward_weight(cp(x), cp(go), cp(gw), None, B, H, W, C, Co, k, k, s, k // 2, 1, 0, st)
        for _ in range(3):
            assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * y.numel() * C * k * k
        print(f'{name:10s} C={C:4d} Co={Co:4d} k={k} s={s} {H:3d}x{W:3d}  P={B * y.shape[2] * y.shape[3]:6d}  {ms:7.4f} ms  {fl / ms / 1e9:7.1f} TF (wgrad)', flush=True)
        continue
    with torch.no_grad():
        for _ in range(3):
            y = conv2d(x, w, None, s, k // 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = conv2d(x, w, None, s, k // 2)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * y.numel() * C * k * k
    print(f'{name:10s} C={C:4d} Co={Co:4d} k={k} s={s} {H:3d}x{W:3d}  P={B * y.shape[2] * y.shape[3]:6d}  {ms:7.4f} ms  {fl / ms / 1e9:7.1f} TF', flush=True)
