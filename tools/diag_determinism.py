"""Diagnostic: repeat the same LSHead training step several times in one process; compare every
dcn_backward result (cloned on-stream, no host sync) and the final gradients run-to-run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_cases as gc, golden_util as gu
from lsnet_amd.ops import get_backend

dev = torch.device('cuda:0')
task = 'bbox'
cl = len(sys.argv) > 1 and sys.argv[1] == 'nhwc'
be = get_backend(torch.zeros(1, device=dev))
orig = be.dcn_backward
calls = []


def rec(inputs, offsets, masks, weight, grad_outs, cfg, need):
    res = orig(inputs, offsets, masks, weight, grad_outs, cfg, need)
    cl_ = lambda t: None if t is None else t.clone()
    calls.append(dict(gx=[cl_(t) for t in res[0]], goff=[cl_(t) for t in res[1]], gw=cl_(res[3]), gb=cl_(res[4]),
                      go=[cl_(t) for t in grad_outs], x=[cl_(t) for t in inputs]))
    return res


be.dcn_backward = rec


def run():
    calls.clear()
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
    g = {f'feat{i}': f.grad.clone() for i, f in enumerate(feats)}
    for n, p in head.named_parameters():
        if p.grad is not None:
            g[n] = p.grad.clone()
    return list(calls), g


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


runs = [run() for _ in range(4)]
torch.cuda.synchronize()
base_calls, base_g = runs[0]
for r, (cs, g) in enumerate(runs[1:], 1):
    bad = [(k, f'{rel(g[k], base_g[k]):.1e}') for k in g if rel(g[k], base_g[k]) > 1e-4]
    print(f'run {r} vs 0: final grads BAD:', bad[:6], '...' if len(bad) > 6 else '')
    for ci, (c, b) in enumerate(zip(cs, base_calls)):
        msgs = []
        for key in ('go', 'x', 'gx', 'goff'):
            for i, (t, u) in enumerate(zip(c[key], b[key])):
                if t is not None and rel(t, u) > 1e-4:
                    msgs.append(f'{key}[{i}] {rel(t, u):.1e}')
        for key in ('gw', 'gb'):
            if c[key] is not None and rel(c[key], b[key]) > 1e-4:
                msgs.append(f'{key} {rel(c[key], b[key]):.1e}')
        if msgs:
            print(f'   call {ci}:', msgs[:8])
