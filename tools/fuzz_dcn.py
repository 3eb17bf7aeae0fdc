"""Randomised parity sweep of the deformable operators against the CPU oracle: multi-level launches with ragged level
sizes, channel counts on both sides of the round-3 kernels' conditions, deformable groups, stride / dilation, masks on
and off, batch sizes 1 - 3.  A diagnostic (not part of the suite): python tools/fuzz_dcn.py [cases] [seed]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lsnet_amd import ops
from oracle import oracle_py as orc

dev = torch.device('cuda:0')
cl = torch.channels_last
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = 1e-4
worst = 0.0


def err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


for it in range(ncase):
    C = int(rng.choice([32, 64, 128, 256, 512]))
    Co = int(rng.choice([64, 128, 256, 512]))
    dg = int(rng.choice([1, 1, 2, 4]))
    if (C // dg) % 4:
        dg = 1
    stride = int(rng.choice([1, 1, 1, 2]))
    dil = int(rng.choice([1, 1, 2]))
    pad = dil
    has_mask = bool(rng.integers(0, 2))
    B = int(rng.integers(1, 4))
    nlv = int(rng.integers(1, 5))
    sizes = [(int(rng.integers(3, 40)), int(rng.integers(3, 40))) for _ in range(nlv)]
    g = torch.Generator().manual_seed(1000 + it)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g) if has_mask else None
    xs, offs, msks, gos, refs = [], [], [], [], []
    gw_ref, gb_ref = torch.zeros_like(w), torch.zeros(Co)
    for (H, W) in sizes:
        Ho, Wo = orc.out_size(H, 3, stride, pad, dil), orc.out_size(W, 3, stride, pad, dil)
        x = torch.randn(B, C, H, W, generator=g)
        off = torch.rand(B, dg * 18, Ho, Wo, generator=g) * 5 - 2.5
        m = torch.rand(B, dg * 9, Ho, Wo, generator=g) if has_mask else None
        go = torch.randn(B, Co, Ho, Wo, generator=g)
        out_ref = orc.deform_conv_forward(x, w, b, off, m, stride, pad, dil, 1, dg, 1.0, 1.0, out_hw=(Ho, Wo))
        gr = orc.deform_conv_backward(x, w, off, m, go, stride, pad, dil, 1, dg, 1.0, 1.0)
        gw_ref += gr['gw']
        if has_mask:
            gb_ref += gr['gb'] if 'gb' in gr else go.sum((0, 2, 3))
        xs.append(x), offs.append(off), msks.append(m), gos.append(go), refs.append((out_ref, gr))
    t = lambda v: None if v is None else v.to(dev).contiguous(memory_format=cl).requires_grad_() if v.dim() == 4 else v.to(dev).requires_grad_()
    xd, od = [t(x) for x in xs], [t(o) for o in offs]
    md = [t(m) for m in msks]
    wd, bd = t(w), (None if b is None else t(b))
    outs = ops.dcn_multi(xd, od, md if has_mask else None, wd, bd, stride, pad, dil, 1, dg)
    wrt = [wd] + ([bd] if bd is not None else []) + xd + od + ([m for m in md] if has_mask else [])
    grads = torch.autograd.grad(outs, wrt, [go.to(dev).contiguous(memory_format=cl) for go in gos])
    e = {}
    e['out'] = max(err(o, r[0]) for o, r in zip(outs, refs))
    k = 0
    e['gw'] = err(grads[k], gw_ref); k += 1
    if bd is not None:
        e['gb'] = err(grads[k], gb_ref); k += 1
    e['gx'] = max(err(grads[k + i], refs[i][1]['gx']) for i in range(nlv)); k += nlv
    e['goff'] = max(err(grads[k + i], refs[i][1]['goff']) for i in range(nlv)); k += nlv
    if has_mask:
        e['gmask'] = max(err(grads[k + i], refs[i][1]['gmask']) for i in range(nlv))
    bad = {n: v for n, v in e.items() if not v < TOL}
    worst = max(worst, max(e.values()))
    print(f'case {it}: C={C} Co={Co} dg={dg} s={stride} d={dil} mask={has_mask} B={B} levels={sizes}', 'OK' if not bad else f'FAIL {bad}',
          flush=True)
print('worst', worst)

# ---- pyramid mode: several offset grids of other resolutions sample ONE source map (shared grad_input buffer) ----
for it in range(max(ncase // 3, 1)):
    C = int(rng.choice([32, 64, 256]))
    Co = int(rng.choice([64, 128, 256]))
    B = int(rng.integers(1, 3))
    nsrc = int(rng.integers(1, 4))
    g = torch.Generator().manual_seed(5000 + it)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    srcs = [torch.randn(B, C, int(rng.integers(4, 30)), int(rng.integers(4, 30)), generator=g) for _ in range(nsrc)]
    pairs = [(int(rng.integers(0, nsrc)), (int(rng.integers(3, 26)), int(rng.integers(3, 26)))) for _ in range(int(rng.integers(2, 7)))]
    offs, gos, scales = [], [], []
    gx_ref = [torch.zeros_like(s) for s in srcs]
    gw_ref = torch.zeros_like(w)
    out_refs, goff_refs = [], []
    for si, (Ho, Wo) in pairs:
        H, W = srcs[si].shape[2:]
        sh, sw = H / Ho, W / Wo
        off = (torch.rand(B, 18, Ho, Wo, generator=g) * 4 - 2) * max(sh, 1.0)
        go = torch.randn(B, Co, Ho, Wo, generator=g)
        out_refs.append(orc.deform_conv_forward(srcs[si], w, None, off, None, 1, 1, 1, 1, 1, sh, sw, out_hw=(Ho, Wo)))
        gr = orc.deform_conv_backward(srcs[si], w, off, None, go, 1, 1, 1, 1, 1, sh, sw)
        gx_ref[si] += gr['gx']
        gw_ref += gr['gw']
        goff_refs.append(gr['goff'])
        offs.append(off), gos.append(go), scales.append((sh, sw))
    sd = [s.to(dev).contiguous(memory_format=cl).requires_grad_() for s in srcs]
    od = [o.to(dev).contiguous(memory_format=cl).requires_grad_() for o in offs]
    wd = w.to(dev).contiguous(memory_format=cl).requires_grad_()
    outs = ops.dcn_multi([sd[si] for si, _ in pairs], od, None, wd, None, 1, 1, 1, scales=scales, pyramid=True)
    used = sorted({si for si, _ in pairs})
    grads = torch.autograd.grad(outs, [wd] + [sd[i] for i in used] + od, [go.to(dev).contiguous(memory_format=cl) for go in gos])
    e = {'out': max(err(o, r) for o, r in zip(outs, out_refs)), 'gw': err(grads[0], gw_ref),
         'gx': max(err(grads[1 + j], gx_ref[i]) for j, i in enumerate(used)),
         'goff': max(err(grads[1 + len(used) + j], goff_refs[j]) for j in range(len(pairs)))}
    bad = {n: v for n, v in e.items() if not v < TOL}
    worst = max(worst, max(e.values()))
    print(f'pyramid {it}: C={C} Co={Co} B={B} sources={[tuple(s.shape[2:]) for s in srcs]} pairs={pairs}', 'OK' if not bad else f'FAIL {bad}', flush=True)
print('worst incl. pyramid', worst)
