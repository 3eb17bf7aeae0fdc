"""Diagnostic: where do grad_offset errors of the split backward kernel sit at large launches?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lsnet_amd import _lib, ops
from oracle import oracle_py as orc

dev = torch.device('cuda:0')
cl = torch.channels_last


def run(H, W, mode, flag, seed=21, off_scale=1.5, B=2, C=256):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(C, generator=g)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * off_scale
    m = torch.rand(B, 9, H, W, generator=g)
    go = torch.randn(B, C, H, W, generator=g)
    _lib.load().lsn_debug_phase_clocks(None, flag)
    _lib.set_math_mode(mode)
    outs = []
    for rep in range(2):
        xd = x.to(dev).contiguous(memory_format=cl).requires_grad_()
        od = off.to(dev).contiguous(memory_format=cl).requires_grad_()
        md = m.to(dev).contiguous(memory_format=cl).requires_grad_()
        wd = w.to(dev).contiguous(memory_format=cl).requires_grad_()
        out = ops.dcn_multi([xd], [od], [md], wd, b.to(dev), 1, 1, 1)[0]
        gx, goff, gm = torch.autograd.grad(out, [xd, od, md], go.to(dev).contiguous(memory_format=cl))
        torch.cuda.synchronize()
        outs.append((gx.cpu(), goff.cpu(), gm.cpu()))
    _lib.load().lsn_debug_phase_clocks(None, 0)
    ref = orc.deform_conv_backward(x, w, off, m, go, 1, 1, 1)
    goff = outs[0][1]
    d = (goff - ref['goff']).abs()
    scale = ref['goff'].abs().max()
    bad = (d > 1e-4 * scale)
    same = torch.equal(outs[0][1], outs[1][1])
    print(f'{H}x{W} {mode} flag {flag:#x}: goff err {d.max() / scale:.2e} frac bad {bad.float().mean():.5f} '
          f'rerun identical {same}; gmask err {(outs[0][2] - ref["gmask"]).abs().max() / ref["gmask"].abs().max():.1e} '
          f'gx err {(outs[0][0] - ref["gx"]).abs().max() / ref["gx"].abs().max():.1e}')
    if bad.any():
        idx = bad.nonzero()          # (b, ch, y, x)
        pix = idx[:, 0] * H * W + idx[:, 2] * W + idx[:, 3]
        tile, pl = pix // 64, pix % 64
        print('  bad entries', len(idx), 'distinct tiles', len(torch.unique(tile)), 'of', (B * H * W + 63) // 64)
        print('  by tap      ', torch.bincount(idx[:, 1] // 2, minlength=9).tolist())
        print('  by dy/dx    ', torch.bincount(idx[:, 1] % 2, minlength=2).tolist())
        print('  by wave     ', torch.bincount(pl // 16, minlength=4).tolist())
        print('  by kq       ', torch.bincount((pl % 16) // 4, minlength=4).tolist())
        print('  by r        ', torch.bincount(pl % 4, minlength=4).tolist())
        t, cnt = torch.unique(tile, return_counts=True)
        print('  entries per bad tile: min/median/max', cnt.min().item(), cnt.median().item(), cnt.max().item(),
              ' first tiles', t[:12].tolist())
        # are the bad values explained by a missing / doubled chunk?  ratio got / ref
        r = (goff[bad] / ref['goff'][bad])
        print('  got/ref quantiles', np.quantile(r.numpy(), [0.05, 0.25, 0.5, 0.75, 0.95]).round(3).tolist())
        # position of the sample: is it near the border?
        ys, xs = idx[:, 2].float(), idx[:, 3].float()
        k = idx[:, 1] // 2
        oy = off[idx[:, 0], 2 * k, idx[:, 2], idx[:, 3]]
        ox = off[idx[:, 0], 2 * k + 1, idx[:, 2], idx[:, 3]]
        py = ys - 1 + (k // 3).float() + oy
        px = xs - 1 + (k % 3).float() + ox
        print('  sample py range', py.min().item(), py.max().item(), 'px range', px.min().item(), px.max().item(),
              ' frac with py<0|py>H-1|px<0|px>W-1:', ((py < 0) | (py > H - 1) | (px < 0) | (px > W - 1)).float().mean().item())
        print('  |offset| of bad samples: mean', oy.abs().mean().item(), ox.abs().mean().item(),
              'vs overall', off.abs().mean().item())


flags = [int(f, 0) for f in os.environ.get('DBG_FLAGS', '0').split(',')]
for f in flags:
    run(100, 168, os.environ.get('DBG_MATH', 'bf16x6'), f)
