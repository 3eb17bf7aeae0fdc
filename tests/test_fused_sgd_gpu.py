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


def test_fused_step_invalidates_the_weight_images():
    """The convolutions read prepared weight images (ops/conv.py); an optimizer that writes the parameters from its own
    kernel has to tell the cache (conv.parameters_updated) -- round 4's first version did not, and every dense convolution
    ran on the weights of step 0 while the benchmark got 0.25 ms faster."""
    import torch.nn.functional as F
    from lsnet_amd.ops.conv import Conv2d
    from lsnet_amd.runner.fused_sgd import ClipSGD
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    conv = Conv2d(64, 128, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(2, 64, 20, 24, device=dev).contiguous(memory_format=torch.channels_last)
    y0 = conv(x)                                                  # builds the image of the initial weight
    w0 = conv.weight.detach().clone()
    conv.weight.grad = torch.randn_like(conv.weight)
    opt = torch.optim.SGD(conv.parameters(), lr=0.5, momentum=0.9, weight_decay=0.0)
    plan = ClipSGD(opt, dict(max_norm=1e9, norm_type=2))
    assert plan.ok
    plan.step()
    assert float((conv.weight.detach() - w0).abs().max()) > 0.1    # the weight moved ...
    y1 = conv(x)
    ref = F.conv2d(x.double(), conv.weight.detach().double(), None, 1, 1)
    assert float((y1.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())   # ... and the convolution saw it
    assert float((y1 - y0).abs().max()) > 0.1


def test_runner_steps_equal_torch_optimizer_steps():
    """Four iterations of the runner on the real detector (data-parallel wrapper: gradients in the reducer's buckets, so the
    hook takes the library path) against the same four with clip_grad_norm_ + torch.optim.SGD.step(): the same losses --
    at the full base learning rate, where a convolution left on stale weights shows in the second iteration."""
    import sys
    import bench
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.parallel import DataParallelModel
    from lsnet_amd.runner import hooks
    dev = torch.device('cuda:0')

    def run(fused):
        torch.manual_seed(0)
        model, cfg = build_lsnet('bbox', 'r50')
        cfg.lr_config = dict(policy='step', step=[8, 11])          # no warm-up: lr = 0.01 from the first iteration
        model = DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
        real = hooks.OptimizerHook._fused_step
        used = []
        try:
            if fused:
                hooks.OptimizerHook._fused_step = lambda self, runner: (used.append(1), real(self, runner))[1]
            else:
                hooks.OptimizerHook._fused_step = lambda self, runner: None
            step, runner = bench.build_step(model, cfg)
            data = synthetic_batch('bbox', 2, 384, 480, seed=5, device=dev)
            losses = []
            for _ in range(4):
                losses.append(float(step(data)['log_vars']['loss']))
        finally:
            hooks.OptimizerHook._fused_step = real
        return losses, len(used)

    ref, _ = run(False)
    got, n = run(True)
    assert n == 4 and getattr(hooks, 'OptimizerHook')
    assert abs(ref[0] - got[0]) < 1e-6 * abs(ref[0])
    assert abs(ref[1] - ref[0]) > 1e-3 * abs(ref[0])               # the steps move the loss ...
    for a, b in zip(ref, got):
        assert abs(a - b) < 2e-4 * abs(a), (ref, got)              # ... and both optimizers move it alike
