"""Diagnostic: LSHead forward/backward in NCHW vs channels_last on the GPU, per-tensor differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_cases as gc, golden_util as gu

dev = torch.device('cuda:0')
task = sys.argv[1] if len(sys.argv) > 1 else 'bbox'


def run(cl):
    head = gc.build_head(task, dev).train()
    feats = [f.to(dev) for f in gu.head_inputs(11)]
    if cl:
        head = head.to(memory_format=torch.channels_last)
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    feats = [f.requires_grad_() for f in feats]
    outs = head(feats)
    boxes, labels, extremes, masks, kps, metas = gc.gt_for(task, dev)
    losses = head.loss(*outs, boxes, extremes if task in ('bbox', 'pose_bbox') else None,
                       [k.clone() for k in kps] if 'pose' in task else None, masks if task == 'segm' else None,
                       labels, metas)
    total = sum(sum(v) for v in losses.values())
    total.backward()
    g = {f'feat{i}': f.grad.detach().cpu().contiguous() for i, f in enumerate(feats)}
    for n, p in head.named_parameters():
        if p.grad is not None:
            g[n] = p.grad.detach().cpu().contiguous()
    o = {}
    for n, lv in zip(['cls', 'bi', 'br'], outs[:3]):
        for i, t in enumerate(lv):
            if t is not None:
                o[f'{n}{i}'] = t.detach().cpu().contiguous()
    return o, g, float(total)


o1, g1, t1 = run(False)
o2, g2, t2 = run(True)
print('total', t1, t2)
for k in o1:
    d = (o1[k] - o2[k]).abs().max().item() / max(o1[k].abs().max().item(), 1e-12)
    if d > 1e-5:
        print('OUT', k, f'{d:.2e}')
for k in g1:
    d = (g1[k] - g2[k]).abs().max().item() / max(g1[k].abs().max().item(), 1e-12)
    flag = 'BAD' if d > 1e-4 else 'ok '
    print(flag, k, f'{d:.2e}', tuple(g1[k].shape))
