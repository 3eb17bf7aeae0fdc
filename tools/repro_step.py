"""Is a training step bit-reproducible?  Builds LSNet R-50 bbox twice from the same seed, runs forward + backward on the
same batch and compares every parameter gradient bitwise; prints the parameters that differ, grouped by module."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet

dev = 'cuda:0'
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 1344)


def run():
    torch.manual_seed(3)
    model, cfg = build_lsnet('bbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last)
    model.train()
    fwd, bwd = [], []

    def fhook(name):
        def f(mod, inp, out):
            ts = out if isinstance(out, (list, tuple)) else [out]
            for i, t in enumerate(ts):
                if torch.is_tensor(t) and t.is_floating_point():
                    fwd.append((f'{name}[{i}]', t.detach().clone()))
                    if t.requires_grad:
                        t.register_hook(lambda g, n=f'{name}[{i}]': bwd.append((n, g.detach().clone())))
        return f
    for n, m in model.named_modules():
        if n.startswith('bbox_head') and len(list(m.children())) == 0:
            m.register_forward_hook(fhook(n))
    data = synthetic_batch('bbox', 2, H, W, boxes_per_img=7, num_classes=80, seed=11, device=dev, channels_last=True)
    losses = model(**data)
    loss = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, fwd, bwd


l1, g1, f1, b1 = run()
l2, g2, f2, b2 = run()
nf = [a[0] for a, b in zip(f1, f2) if not torch.equal(a[1], b[1])]
print(len(nf), 'of', len(f1), 'forward outputs of head modules differ; first:', nf[:4])
nb = [(a[0], float((a[1] - b[1]).abs().max() / a[1].abs().max().clamp_min(1e-30))) for a, b in zip(b1, b2) if not torch.equal(a[1], b[1])]
print(len(nb), 'of', len(b1), 'gradients w.r.t. head module outputs differ; in backward order:')
seen = set(x[0] for x in nb)
for a in b1[:60]:
    print('   ', 'DIFF ' if a[0] in seen else 'equal', a[0], tuple(a[1].shape))
print('loss', l1, l2, 'equal' if l1 == l2 else 'DIFFERENT')
diff = [n for n in g1 if not torch.equal(g1[n], g2[n])]
print(len(diff), 'of', len(g1), 'parameter gradients differ')
groups = {}
for n in diff:
    rel = float((g1[n] - g2[n]).abs().max() / g1[n].abs().max().clamp_min(1e-30))
    groups.setdefault('.'.join(n.split('.')[:2]), []).append((n, rel))
for k, v in groups.items():
    print(' ', k, len(v), 'e.g.', v[0][0], f'{v[0][1]:.1e}')
print('head / neck parameters, in registration order:')
for n in g1:
    if n.startswith('bbox_head') or n.startswith('neck'):
        print('   ', 'DIFF ' if n in diff else 'equal', n, tuple(g1[n].shape))
