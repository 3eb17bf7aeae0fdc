"""Diagnostic: record every dcn_backward call of an LSHead training step in channels_last mode and
re-check each against the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_cases as gc, golden_util as gu
from lsnet_amd.ops import get_backend
from oracle import oracle_py as orc

dev = torch.device('cuda:0')
task = 'bbox'
cl = len(sys.argv) < 2 or sys.argv[1] != 'nchw'
be = get_backend(torch.zeros(1, device=dev))
calls = []
orig = be.dcn_backward


def rec(inputs, offsets, masks, weight, grad_outs, cfg, need):
    res = orig(inputs, offsets, masks, weight, grad_outs, cfg, need)
    torch.cuda.synchronize()
    cp = lambda t: None if t is None else t.detach().cpu().contiguous().clone()
    calls.append(dict(x=[cp(t) for t in inputs], off=[cp(t) for t in offsets], m=[cp(t) for t in masks], w=cp(weight),
                      go=[cp(t) for t in grad_outs], cfg=dict(cfg), gx=[cp(t) for t in res[0]],
                      goff=[cp(t) for t in res[1]], gm=[cp(t) for t in res[2]], gw=cp(res[3]), gb=cp(res[4]),
                      strides=[(tuple(t.stride()), tuple(g.stride())) for t, g in zip(inputs, grad_outs)]))
    return res


be.dcn_backward = rec
head = gc.build_head(task, dev).train()
feats = [f.to(dev) for f in gu.head_inputs(11)]
if cl:
    head = head.to(memory_format=torch.channels_last)
    feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
feats = [f.requires_grad_() for f in feats]
outs = head(feats)
boxes, labels, extremes, masks, kps, metas = gc.gt_for(task, dev)
losses = head.loss(*outs, boxes, extremes, None, None, labels, metas)
sum(sum(v) for v in losses.values()).backward()
print('recorded', len(calls), 'backward calls; channels_last =', cl)
for ci, c in enumerate(calls):
    cfg = c['cfg']
    gw = torch.zeros_like(c['w']); worst = {}
    for i in range(len(c['x'])):
        sh, sw = cfg['scales'][i]
        g = orc.deform_conv_backward(c['x'][i], c['w'], c['off'][i], c['m'][i], c['go'][i], cfg['stride'], cfg['pad'],
                                     cfg['dil'], cfg['groups'], cfg['dg'], sh, sw)
        gw += g['gw']
        for name, got, ref in (('gx', c['gx'][i], g['gx']), ('goff', c['goff'][i], g['goff']), ('gm', c['gm'][i], g['gmask'])):
            if got is None or ref is None:
                continue
            e = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            if e > 1e-4:
                worst[f'{name}[{i}]'] = (f'{e:.1e}', tuple(c['x'][i].shape), c['strides'][i])
    e = (c['gw'] - gw).abs().max().item() / max(gw.abs().max().item(), 1e-12)
    if e > 1e-4:
        worst['gw'] = f'{e:.1e}'
    print('call', ci, 'levels', len(c['x']), 'pyramid' if cfg['pyramid'] else 'dcnv2', 'BAD ' + str(worst) if worst else 'ok')
