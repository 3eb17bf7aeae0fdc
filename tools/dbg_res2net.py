"""Where does the Res2Net-50-DCN backward on the device leave the host (oracle-backed) one?  Same weights, same input: every
parameter gradient and the gradient of every block output, in backward order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
from lsnet_amd.models import build_backbone
from lsnet_amd.ops import register_backend
from tests import golden_util as gu
from tests.oracle_backend import OracleBackend

register_backend('cpu', OracleBackend())
cfg = dict(type='Res2Net', depth=50, scales=4, base_width=26, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
           norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
           dcn=dict(type='DCNv2', deformable_groups=1, fallback_on_stride=False), stage_with_dcn=(False, True, True, True))
bb = build_backbone(cfg)
gu.fill_params(bb, seed=8)
bb.train()
dev = torch.device('cuda:0')
bd = copy.deepcopy(bb).to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(1, 3, 96, 128, generator=gu.gen(31))


def run(model, x):
    grads = {}
    def hook(name):
        def f(mod, inp, out):
            if torch.is_tensor(out) and out.requires_grad:
                out.register_hook(lambda g, n=name: grads.__setitem__('act:' + n, g.detach().float().cpu().contiguous()))
        return f
    hs = [m.register_forward_hook(hook(n)) for n, m in model.named_modules() if n and (n.count('.') <= 2 or n.startswith('layer4.0') or n.startswith('layer4.1'))]
    x = x.clone().requires_grad_()
    feats = model(x)
    proj = sum((f * torch.randn(f.shape, generator=gu.gen(60 + i)).to(f.device)).sum() / f.numel() ** 0.5 for i, f in enumerate(feats))
    proj.backward()
    for h in hs:
        h.remove()
    grads['gx'] = x.grad.detach().float().cpu()
    for n, p in model.named_parameters():
        if p.grad is not None:
            grads['par:' + n] = p.grad.detach().float().cpu().contiguous()
    return [f.detach().float().cpu() for f in feats], grads

fc, gc_ = run(bb, x)
fd, gd = run(bd, x.to(dev).contiguous(memory_format=torch.channels_last))
for i, (a, b) in enumerate(zip(fc, fd)):
    print('feat', i, float((a - b).abs().max() / a.abs().max()))
rows = []
for k in gc_:
    if k in gd:
        a, b = gc_[k], gd[k]
        rows.append((float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)), k, tuple(a.shape)))
bad = [r for r in rows if r[0] > 1e-3]
print(len(bad), 'of', len(rows), 'tensors off by more than 1e-3')
order = list(gc_.keys())
for r in sorted(rows, key=lambda r: order.index(r[1])):
    if (r[1].startswith('act:') and 'layer4' in r[1]) or (r[0] > 1e-3 and 'layer4' in r[1]):
        print(f'{r[0]:9.2e}  {r[1]}  {r[2]}')
a, b = gc_['act:layer4.0.bn1'], gd['act:layer4.0.bn1']
for q in range(4):
    sa, sb = a[:, q * 208:(q + 1) * 208], b[:, q * 208:(q + 1) * 208]
    print('layer4.0.bn1 grad, scale', q, float((sa - sb).abs().max() / sa.abs().max()), 'host absmax', float(sa.abs().max()), 'device absmax', float(sb.abs().max()))
