"""runner/fused_sgd.py without a GPU: the plan declines everything the library's clip + SGD step does not do (the hook then
keeps clip_grad_norm_ + optimizer.step(), mmcv/runner/hooks/optimizer.py:8-28), and the host-side fall-backs of the round-4
glue (core/assigners.py: topk_columns, ops/dcn.py: offset_scale_chain) are the torch statements they replace."""
import torch

from lsnet_amd.core.assigners import topk_columns
from lsnet_amd.ops.dcn import offset_scale_chain
from lsnet_amd.runner.fused_sgd import ClipSGD


def _params():
    ps = [torch.nn.Parameter(torch.randn(8, 4, 3, 3)), torch.nn.Parameter(torch.randn(8))]
    for p in ps:
        p.grad = torch.randn_like(p)
    return ps


def test_plan_declines_what_the_library_does_not_do():
    clip = dict(max_norm=35, norm_type=2)
    ps = _params()
    assert not ClipSGD(torch.optim.SGD(ps, lr=0.1, momentum=0.9), clip).ok                     # host tensors
    assert not ClipSGD(torch.optim.Adam(ps, lr=0.1), clip).ok                                  # not SGD
    assert not ClipSGD(torch.optim.SGD(ps, lr=0.1, momentum=0.9, nesterov=True), clip).ok
    assert not ClipSGD(torch.optim.SGD(ps, lr=0.1, momentum=0.0), clip).ok                     # no momentum buffer to keep
    assert not ClipSGD(torch.optim.SGD(ps, lr=0.1, momentum=0.9), dict(max_norm=35, norm_type=1)).ok
    assert not ClipSGD(torch.optim.SGD(ps, lr=0.1, momentum=0.9), dict(max_norm=0.0, norm_type=2)).ok


def test_topk_columns_on_the_host_is_torch_topk():
    g = torch.Generator().manual_seed(0)
    x = torch.randperm(400 * 5, generator=g).float().reshape(400, 5)
    v, i = topk_columns(x, 3)
    vr, ir = x.topk(3, dim=0, largest=False)
    assert torch.equal(v, vr) and torch.equal(i, ir)
    segs = [(0, 250), (250, 100), (350, 50)]
    v, i = topk_columns(x, 4, segs, largest=True)
    for s, (start, n) in enumerate(segs):
        vr, ir = x[start:start + n].topk(4, dim=0, largest=True)
        assert torch.equal(v[4 * s:4 * s + 4], vr) and torch.equal(i[4 * s:4 * s + 4], ir + start)


def test_offset_scale_chain_on_the_host_is_the_multiplication_sequence():
    g = torch.Generator().manual_seed(1)
    offs = [torch.randn(2, 18, 5, 7, generator=g, requires_grad=True), torch.randn(2, 18, 3, 4, generator=g, requires_grad=True)]
    mults = [((0.5, 0.52), (2.0, 25 / 13), (0.28, 11 / 42)), ((1.0, 1.0), (13 / 7, 21 / 11), (7 / 13, 0.5))]
    first, second = offset_scale_chain(offs, mults, copies=2)
    assert first is second or all(a is b for ta, tb in zip(first, second) for a, b in zip(ta, tb))   # host: the same tensors
    for off, m, trio in zip(offs, mults, first):
        cur = off
        for (sh, sw), t in zip(m, trio):
            cur = cur * off.new_tensor([sh, sw]).repeat(9).view(1, -1, 1, 1)
            assert torch.equal(t, cur)
    sum(t.sum() for trio in first for t in trio).backward()
    assert all(o.grad is not None for o in offs)
