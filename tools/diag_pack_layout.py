"""Diagnostic: a ModulatedDeformConvPack tower in NCHW vs channels_last, and run-to-run determinism."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_util as gu
from lsnet_amd.models.dense_heads.ls_head import DCNConvModule

dev = torch.device('cuda:0')
C = 32
sizes = [(48, 64), (24, 32), (12, 16), (6, 8), (3, 4)]


def run(cl, nlayers=3, multi=True, torch_offset_only=False):
    torch.manual_seed(0)
    tower = torch.nn.ModuleList([DCNConvModule(C, C, 3, 1, 8, 1) for _ in range(nlayers)])
    gu.fill_params(tower, seed=5)
    tower = tower.to(dev)
    xs = [torch.randn(2, C, h, w, generator=gu.gen(9 + i)).to(dev) for i, (h, w) in enumerate(sizes)]
    if cl:
        tower = tower.to(memory_format=torch.channels_last)
        xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    xs = [x.requires_grad_() for x in xs]
    out = xs
    for m in tower:
        out = m.forward_multi(out) if multi else [m(x) for x in out]
    loss = sum((o * o).sum() for o in out)
    loss.backward()
    g = {f'x{i}': x.grad.detach().cpu().contiguous() for i, x in enumerate(xs)}
    for n, p in tower.named_parameters():
        g[n] = p.grad.detach().cpu().contiguous()
    return g


def cmp(a, b, tag):
    bad = []
    for k in a:
        d = (a[k] - b[k]).abs().max().item() / max(a[k].abs().max().item(), 1e-12)
        if d > 1e-4:
            bad.append((k, f'{d:.1e}'))
    print(tag, 'BAD:' if bad else 'all ok', bad)


for nl in (1, 2, 3):
    for multi in (True, False):
        a = run(False, nl, multi)
        b = run(True, nl, multi)
        c = run(True, nl, multi)
        cmp(a, b, f'layers={nl} multi={multi} nchw-vs-nhwc')
        cmp(b, c, f'layers={nl} multi={multi} nhwc-vs-nhwc')
