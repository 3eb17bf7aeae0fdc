"""Diagnostic: does any native call corrupt a live tensor?  Snapshot every DCN output right after
it is produced and re-verify all snapshots after every later native call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_cases as gc, golden_util as gu
from lsnet_amd.ops import get_backend

dev = torch.device('cuda:0')
task = 'bbox'
cl = len(sys.argv) > 1 and sys.argv[1] == 'nhwc'
be = get_backend(torch.zeros(1, device=dev))
live = []   # (name, tensor, snapshot)
ncall = [0]


def verify(tag):
    torch.cuda.synchronize()
    for name, t, snap in live:
        if not torch.equal(t, snap):
            d = (t - snap).abs()
            print(f'CORRUPTED after {tag}: {name} shape {tuple(t.shape)} strides {t.stride()} '
                  f'n_bad {(d > 0).sum().item()} first_bad_flat {torch.nonzero(d.flatten() > 0)[0].item()}')
            live[:] = [(n, a, a.clone()) if n == name else (n, a, s) for n, a, s in live]


of, ob = be.dcn_forward, be.dcn_backward


def fwd(inputs, offsets, masks, weight, bias, cfg, out_hw):
    outs = of(inputs, offsets, masks, weight, bias, cfg, out_hw)
    ncall[0] += 1
    verify(f'fwd call {ncall[0]} ({len(inputs)} levels, pyramid={cfg["pyramid"]})')
    for i, o in enumerate(outs):
        live.append((f'out of fwd call {ncall[0]} level {i}', o, o.clone()))
    for i, x in enumerate(inputs):
        live.append((f'input of fwd call {ncall[0]} level {i}', x, x.clone()))
    return outs


def bwd(inputs, offsets, masks, weight, grad_outs, cfg, need):
    res = ob(inputs, offsets, masks, weight, grad_outs, cfg, need)
    ncall[0] += 1
    verify(f'bwd call {ncall[0]} ({len(inputs)} levels, pyramid={cfg["pyramid"]})')
    return res


be.dcn_forward, be.dcn_backward = fwd, bwd
for rep in range(2):
    live.clear()
    head = gc.build_head(task, dev).train()
    feats = [f.to(dev) for f in gu.head_inputs(11)]
    if cl:
        head = head.to(memory_format=torch.channels_last)
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    feats = [f.requires_grad_() for f in feats]
    outs = head(feats)
    verify('forward done')
    boxes, labels, extremes, masks, kps, metas = gc.gt_for(task, dev)
    losses = head.loss(*outs, boxes, extremes, None, None, labels, metas)
    verify('loss done')
    sum(sum(v) for v in losses.values()).backward()
    verify('backward done')
    print('rep', rep, 'done; tracked', len(live))
