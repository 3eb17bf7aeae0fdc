"""Own conv kernels (forward / data gradient / weight gradient, current math mode) vs MIOpen on the conv shapes of the
R-50 benchmark step (B = 2, 800x1344), per shape: where the time of the vendor-free step goes."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lsnet_amd import _lib
from lsnet_amd.ops.conv import conv2d
from tools.bench_convs import SH, B

dev = torch.device('cuda:0')
mode = os.environ.get('LSNET_MATH', 'bf16x6')
OWN_ONLY = '--own-only' in sys.argv   # skip the MIOpen columns (tile sweeps: LSNET_CONV_TILE=1|2|3)
_lib.set_math_mode(mode)
EXTRA = [('head 1x1 256->80 P3 (cls out)', 256, 80, 1, 1, 100, 168, 1), ('head 1x1 256->20 P3', 256, 20, 1, 1, 100, 168, 2),
         ('head 3x3 256->27 P4 (offset conv)', 256, 27, 3, 1, 50, 84, 6), ('fpn P6 3x3 s2 2048->256', 2048, 256, 3, 2, 25, 42, 1),
         ('stem 7x7 s2 3->64', 3, 64, 7, 2, 800, 1344, 1)]


def ev_time(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


lib = _lib.load()
cp = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
from lsnet_amd.ops.conv import weight_image, _levels
print(f'mode {mode}; ms per call (own | MIOpen); err = max |own - fp64 reference| / max |reference|')
print(f'{"shape":38s} {"fwd":>15s} {"bwd-data":>15s} {"wgrad":>15s}   GFLOP  xcount   TF(fwd) TB/s(fwd)  err fwd / bwd')
tot = [0.0] * 6
ONLY = os.environ.get('CONV_ONLY')   # substring filter on the shape name
for name, ci, co, k, s, h, w, cnt in SH + EXTRA:
    if ONLY and ONLY not in name:
        continue
    pad = k // 2
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = F.conv2d(x, wt, None, s, pad)
        own = conv2d(x, wt, None, s, pad)
        r64 = F.conv2d(x[:1, :, :min(h, 64)].double(), wt.double(), None, s, pad)
        nr = r64.shape[2] - (4 if h > 64 else 0)   # (the cropped input has a padded bottom edge the full one lacks)
        err_f = float((own[:1, :, :nr].double() - r64[:, :, :nr]).abs().max() / r64.abs().max())
        t_f = ev_time(lambda: conv2d(x, wt, None, s, pad))
        t_fm = 0.0 if OWN_ONLY else ev_time(lambda: F.conv2d(x, wt, None, s, pad))
    fl = 2.0 * ref.numel() * ci * k * k / 1e9
    by = 4.0 * (x.numel() + ref.numel() + wt.numel()) / 1e9
    go = torch.randn_like(ref)
    line = f'{name:38s} {t_f:7.3f}|{t_fm:7.3f}'
    err_d = err_w = 0.0
    if ci >= 8:
        co8 = (co + 3) // 4 * 4
        go8 = go if co8 == co else torch.cat([go, go.new_zeros(B, co8 - co, *go.shape[2:])], 1).contiguous(memory_format=torch.channels_last)
        w8 = wt if co8 == co else torch.cat([wt, wt.new_zeros(co8 - co, ci, k, k)], 0).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x)
        gw = torch.empty_like(wt)
        lv = _levels(1)
        lv[0].x, lv[0].out, lv[0].B, lv[0].H, lv[0].W = cp(go8), cp(gx), B, h, w
        img = weight_image(w8, 1, s, pad, 1)
        f_d = lambda: lib.lsn_conv2d_backward_data_prepared(1, lv, cp(img), ci, co8, k, k, s, pad, 1, st)
        assert f_d() == 0, lib.lsn_last_error()
        gref = torch.ops.aten.convolution_backward(go, x, wt, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        err_d = float((gx - gref).abs().max() / gref.abs().max())
        t_d = ev_time(f_d)
        wref = torch.ops.aten.convolution_backward(go, x, wt, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        assert lib.lsn_conv2d_backward_weight(cp(x), cp(go), cp(gw), None, B, h, w, ci, co, k, k, s, pad, 1, 0, st) == 0, lib.lsn_last_error()
        err_w = float((gw - wref).abs().max() / wref.abs().max())
        t_w = ev_time(lambda: lib.lsn_conv2d_backward_weight(cp(x), cp(go), cp(gw), None, B, h, w, ci, co, k, k, s, pad, 1, 0, st))
        t_dm = 0.0 if OWN_ONLY else ev_time(lambda: torch.ops.aten.convolution_backward(go, x, wt, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
        t_wm = 0.0 if OWN_ONLY else ev_time(lambda: torch.ops.aten.convolution_backward(go, x, wt, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
        line += f' {t_d:7.3f}|{t_dm:7.3f} {t_w:7.3f}|{t_wm:7.3f}'
        for i, v in enumerate((t_f, t_fm, t_d, t_dm, t_w, t_wm)):
            tot[i] += v * cnt
    else:
        tot[0] += t_f * cnt
        tot[1] += t_fm * cnt
        line += ' ' * 32
    print(line + f' {fl:7.1f}  x{cnt}   {fl / t_f:7.1f} {by / t_f:7.2f}   {err_f:.1e} {err_d:.1e} {err_w:.1e}', flush=True)
print(f'network sums (ms): fwd own {tot[0]:.2f} | MIOpen {tot[1]:.2f};  bwd-data own {tot[2]:.2f} | {tot[3]:.2f};  '
      f'wgrad own {tot[4]:.2f} | {tot[5]:.2f}')
