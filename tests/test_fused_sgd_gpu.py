"""lsn_clip_sgd_step (runner/fused_sgd.py) against the operator sequence it replaces -- torch.nn.utils.clip_grad_norm_ and
torch.optim.SGD.step() (mmcv/runner/hooks/optimizer.py:8-28): without clipping the parameters and momentum buffers carry the
SAME BITS as torch's after several steps; with clipping they differ by the rounding of the norm (a different summation order)
and nothing else."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tensors(dev, gen):
    shapes = [(64, 32, 3, 3), (256,), (7,), (5, 3, 1, 1), (1000, 13), (2048, 512, 1, 1), (33,)]
    ps = []
    for s in shapes:
        p = torch.randn(s, generator=gen).to(dev)
        if len(s) == 4:
            p = p.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(p))
    return ps


@pytest.mark.parametrize('max_norm', [1e9, 3.0, None])
def test_clip_sgd_step_matches_torch(max_norm):
    from lsnet_amd.runner.fused_sgd import ClipSGD
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(7)
    mine = _tensors(dev, gen)
    refs = {kind: [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in mine] for kind in ('fused', 'foreach')}
    kw = dict(lr=0.02, momentum=0.9, weight_decay=1e-4)
    opt = torch.optim.SGD(mine, **kw)
    ropt = {'fused': torch.optim.SGD(refs['fused'], fused=True, **kw), 'foreach': torch.optim.SGD(refs['foreach'], foreach=True, **kw)}
    for p in mine:                      # stable gradient storage, as the all-reduce buckets give
        p.grad = torch.zeros_like(p)
    clip = None if max_norm is None else dict(max_norm=max_norm, norm_type=2)
    plan = ClipSGD(opt, clip)
    assert plan.ok
    for step in range(4):
        gs = [torch.randn(p.shape, generator=gen).to(dev) * (10.0 if step == 2 else 0.1) for p in mine]
        for p, g in zip(mine, gs):
            p.grad.copy_(g)
        for kind in refs:
            for p, g in zip(refs[kind], gs):
                p.grad = g.clone().contiguous(memory_format=torch.channels_last) if g.dim() == 4 else g.clone()
        for g in opt.param_groups:      # a schedule: the learning rate changes every step
            g['lr'] = 0.02 * (step + 1)
        assert plan.still_valid()
        norm = plan.step()
        rnorm = {}
        for kind in refs:
            for g in ropt[kind].param_groups:
                g['lr'] = 0.02 * (step + 1)
            if clip is not None:
                rnorm[kind] = torch.nn.utils.clip_grad_norm_(refs[kind], **clip)
            ropt[kind].step()
        if clip is not None:
            assert abs(float(norm) - float(rnorm['foreach'])) <= 2e-6 * float(rnorm['foreach'])
        clipped = clip is not None and float(rnorm['foreach']) > max_norm
        exact = {kind: all(torch.equal(a, b) and torch.equal(opt.state[a]['momentum_buffer'], ropt[kind].state[b]['momentum_buffer'])
                           for a, b in zip(mine, refs[kind])) for kind in refs}
        if not clipped and step < 2 and (max_norm is None or max_norm > 1e6):
            assert exact['foreach'] or exact['fused'], (step, exact)
        for kind in refs:
            for a, b in zip(mine, refs[kind]):
                assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max()) + 1e-9, (kind, step)
                if clipped:         # the gradients were scaled in place, as clip_grad_norm_ leaves them
                    assert float((a.grad - b.grad).abs().max()) <= 2e-6 * float(b.grad.abs().max())
