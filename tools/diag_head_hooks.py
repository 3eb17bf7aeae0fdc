"""Diagnostic: per-module backward hooks on the cls tower; GPU (2nd run in process) vs CPU-oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_cases as gc, golden_util as gu
from lsnet_amd.ops import register_backend
from tests.oracle_backend import OracleBackend

register_backend('cpu', OracleBackend())
task = 'bbox'


def run(dev, cl):
    global _dev_

    head = gc.build_head(task, dev).train()
    feats = [f.to(dev) for f in gu.head_inputs(11)]
    if cl:
        head = head.to(memory_format=torch.channels_last)
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    feats = [f.requires_grad_() for f in feats]
    rec = {}

    from lsnet_amd.models.dense_heads.ls_head import DCNConvModule

    def save(name):
        def hook(g):
            n = 0
            while f'{name}#{n}' in rec:
                n += 1
            rec[f'{name}#{n}'] = dict(gin=[g.detach().cpu().contiguous().clone()], gout=[])
        return hook

    def patched(self, xs):
        tag = self._tag
        oms = [self.conv.conv_offset(x) for x in xs]
        for i, (x, om) in enumerate(zip(xs, oms)):
            if x.requires_grad:
                x.register_hook(save(f'{tag}.x{i}'))
            om.register_hook(save(f'{tag}.om{i}'))
        from lsnet_amd.ops import dcn_multi
        c = self.conv
        ys = dcn_multi(list(xs), oms, None, c.weight, c.bias, c.stride, c.padding, c.dilation, c.groups,
                       c.deformable_groups, fused_om=True)
        outs = []
        for i, y in enumerate(ys):
            y.register_hook(save(f'{tag}.y{i}'))
            outs.append(self.relu(self.bn(y)))
        if dev.type == 'cuda':
            import torch.nn.functional as F
            mod = self
            for i, (y, o) in enumerate(zip(ys, outs)):
                def chk(g, y=y, o=o, i=i):
                    # g = grad wrt y (GN input).  Recompute on CPU from the live tensors.
                    gx2 = rec.get(f'next_x_grad.{tag}.{i}')
                    yc = y.detach().cpu().contiguous().requires_grad_()
                    with torch.enable_grad():
                      oc = F.relu(F.group_norm(yc, mod.bn.num_groups, mod.bn.weight.detach().cpu(), mod.bn.bias.detach().cpu(), mod.bn.eps))
                    live_o = o.detach().cpu().contiguous()
                    e_o = (live_o - oc.detach()).abs().max().item() / oc.abs().max().item()
                    msg = f'[{tag} lvl {i}] live relu(GN(y)) vs recomputed: {e_o:.1e}'
                    if gx2 is not None:
                        oc.backward(gx2)
                        e_g = (g.detach().cpu().contiguous() - yc.grad).abs().max().item() / yc.grad.abs().max().item()
                        msg += f'; torch GN+ReLU backward vs CPU recompute: {e_g:.1e}; g strides {g.stride()} y strides {y.stride()} o strides {o.stride()}'
                    print(msg)
                y.register_hook(chk)
                def keep(g, i=i):
                    rec[f'next_x_grad.{tag}.{i}'] = g.detach().cpu().contiguous().clone()
                o.register_hook(keep)
        return outs
    for tower in ('cls_convs', 'bbox_convs'):
        for i, m in enumerate(getattr(head, tower)):
            m._tag = f'{tower}.{i}'
            m.forward_multi = patched.__get__(m, DCNConvModule)
    outs = head(feats)
    boxes, labels, extremes, masks, kps, metas = gc.gt_for(task, dev)
    losses = head.loss(*outs, boxes, extremes, None, None, labels, metas)
    sum(sum(v) for v in losses.values()).backward()
    rec['feat0'] = dict(gin=[feats[0].grad.detach().cpu().contiguous()], gout=[])
    return rec


gpu = torch.device('cuda:0')
_ = run(gpu, False)           # warm the process (first run is always fine)
bad = run(gpu, len(sys.argv) > 1 and sys.argv[1] == 'nhwc')
ref = run(torch.device('cpu'), False)
for k in ref:
    if k.startswith('next_x_grad'):
        continue
    for kind in ('gout', 'gin'):
        for j, (a, b) in enumerate(zip(bad[k][kind], ref[k][kind])):
            if a is None or b is None:
                continue
            e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
            print('BAD' if e > 1e-4 else 'ok ', k, kind, j, tuple(b.shape), f'{e:.1e}')
            if e > 1e-4 and b.dim() == 4:
                d = (a - b).abs()
                thr = 1e-4 * b.abs().max()
                badmask = d > thr
                nb = badmask.sum().item()
                idx = torch.nonzero(badmask)
                print('     n_bad', nb, 'of', b.numel(), '| batch', idx[:, 0].unique().tolist(), '| chans',
                      idx[:, 1].unique().tolist()[:40], '| rows', idx[:, 2].min().item(), '-', idx[:, 2].max().item(),
                      '| cols', idx[:, 3].min().item(), '-', idx[:, 3].max().item(), '| ref absmax', b.abs().max().item(),
                      'got absmax', a.abs().max().item())
