"""The fused bottleneck node (lsnet_amd/ops/resblock.py) on the MI355X: its four kernel primitives against torch in fp64,
and a whole stage (ResLayer of three bottlenecks: projection shortcut, strided conv2, pregate flags between the blocks)
against autograd over the plain modules in fp64 on the host -- output, input gradient and every parameter gradient,
zero and tiny gammas included."""
import copy

import pytest
import torch

from tests.test_resblock import TorchPrims

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _err(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('C,Co,k,s,case', [(64, 128, 1, 1, 'res+gate'), (128, 128, 3, 2, 'gate'), (256, 64, 1, 1, 'inplace'),
                                           (64, 256, 3, 1, 'plain')])
def test_dgrad_epilogue_and_wgrad_bn(C, Co, k, s, case):
    """dgrad with the folded norm's scale in the image + residual / gate / in-place accumulate in the epilogue, and
    wgrad_bn, against the torch statements of tests/test_resblock.py evaluated in fp64."""
    from lsnet_amd.ops import conv as K
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    B, H, W = 2, 22, 18
    conv = K.Conv2d(C, Co, k, stride=s, padding=k // 2, bias=False).to(dev).to(memory_format=CL)
    bn = torch.nn.BatchNorm2d(Co).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Co) + 0.5)
        bn.weight[:4] = 0.0
        bn.weight[4:8] = 1e-6
        bn.bias.copy_(torch.randn(Co) * 0.3)
        bn.running_mean.copy_(torch.randn(Co) * 0.5)
        bn.running_var.copy_(torch.rand(Co) + 0.5)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=CL)
    g = torch.randn(B, Co, Ho, Wo, device=dev).contiguous(memory_format=CL)
    res = torch.randn_like(x) if case in ('res+gate', 'inplace') else None
    gate = torch.randn_like(x) if case in ('res+gate', 'gate') else None
    out = res.clone() if case == 'inplace' else None
    got = K.dgrad(g, conv.weight, x.shape, s, k // 2, 1, bn=bn, residual=(out if case == 'inplace' else res), gate=gate, out=out)
    d = lambda t: None if t is None else t.detach().double().cpu()
    conv64, bn64 = copy.deepcopy(conv).double().cpu(), copy.deepcopy(bn).double().cpu()
    ref = TorchPrims.dgrad(d(g), conv64.weight, x.shape, s, k // 2, 1, bn=bn64, residual=d(res), gate=d(gate))
    assert _err(got, ref) < 5e-6, _err(got, ref)
    if gate is not None:
        assert bool(((got == 0) | (gate > 0)).all())
    gw, dg, db = K.wgrad_bn(x, g, conv.weight, bn, s, k // 2, 1)
    rw, rg, rb = TorchPrims.wgrad_bn(d(x), d(g), conv64.weight, bn64, s, k // 2, 1)
    for a, r, n in ((gw, rw, 'gw'), (dg, rg, 'dgamma'), (db, rb, 'dbeta')):
        assert _err(a, r) < 5e-6, (n, _err(a, r))
    gate_y = torch.randn_like(g)
    assert torch.equal(K.relu_gate(g, gate_y), g * (gate_y > 0))


@pytest.mark.parametrize('inplanes,planes,n,stride', [(64, 32, 3, 2), (256, 64, 2, 1)])
def test_fused_stage_equals_fp64_autograd(inplanes, planes, n, stride):
    from lsnet_amd.models.backbones import resnet as R
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    layer = R.ResLayer(R.Bottleneck, inplanes, planes, n, stride=stride)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(-1.0, 1.5)
            m.bias.data.normal_(0, 0.3)
            m.running_mean.normal_(0, 0.5)
            m.running_var.uniform_(0.5, 2.0)
    layer[-1].norm3.weight.data[:5] = 0.0
    layer[-1].norm3.weight.data[5:9] = 1e-6
    ref = copy.deepcopy(layer).double()
    layer = layer.to(dev).to(memory_format=CL).train()
    for L in (layer, ref):
        L.train()
        for m in L.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    x = torch.randn(2, inplanes, 40, 36)
    xd = x.to(dev).contiguous(memory_format=CL).requires_grad_()
    assert all(R.fused_block_ok(b, xd) for b in layer)
    y = layer(xd)
    go = torch.randn(y.shape)
    y.backward(go.to(dev))
    xr = x.double().requires_grad_()
    yr = xr
    for b in ref:
        yr = b._body(yr)       # CPU tensors: the plain operator sequence
    yr.backward(go.double())
    assert _err(y, yr) < 1e-5
    assert _err(xd.grad, xr.grad) < 2e-5, _err(xd.grad, xr.grad)
    pr = dict(ref.named_parameters())
    for k, p in layer.named_parameters():
        assert p.grad is not None, k
        assert _err(p.grad, pr[k].grad) < 2e-5, (k, _err(p.grad, pr[k].grad))
    # the zero-initialised norm3 of the last block learns
    assert float(layer[-1].norm3.weight.grad[:5].abs().min()) > 0


def test_identical_blocks_batched_weight_gradients():
    """With gradient sinks registered the blocks of a stage queue their weight gradients and block 0's backward launches
    the layers of one geometry together (lsn_conv2d_backward_weight_bn_jobs): against fp64 autograd on the
    host, every parameter gradient -- at a size where the jobs need pixel splits, accumulated ONTO a non-zero sink."""
    from lsnet_amd.models.backbones import resnet as R
    from lsnet_amd.ops import grad_sink, resblock
    dev = torch.device('cuda:0')
    torch.manual_seed(4)
    layer = R.ResLayer(R.Bottleneck, 128, 64, 4, stride=2)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(-1.0, 1.5)
            m.bias.data.normal_(0, 0.3)
            m.running_mean.normal_(0, 0.5)
            m.running_var.uniform_(0.5, 2.0)
    ref = copy.deepcopy(layer).double()
    layer = layer.to(dev).to(memory_format=CL).train()
    for L in (layer, ref):
        L.train()
        for m in L.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    seed = {}
    for k, p in layer.named_parameters():        # the sinks: p.grad itself, holding something already
        p.grad = torch.full_like(p, 0.25)
        seed[k] = 0.25
        grad_sink.register(p, p.grad)
    launches = []
    real = resblock.K.wgrad_bn_jobs
    try:
        resblock.K.wgrad_bn_jobs = lambda jobs, *cfg: (launches.append(len(jobs)), real(jobs, *cfg))[1]
        x = torch.randn(2, 128, 96, 80)
        xd = x.to(dev).contiguous(memory_format=CL).requires_grad_()
        y = layer(xd)
        go = torch.randn(y.shape)
        y.backward(go.to(dev))
    finally:
        resblock.K.wgrad_bn_jobs = real
        for p in layer.parameters():
            grad_sink.unregister(p)
    assert launches == [4, 3, 3]
    xr = x.double().requires_grad_()
    yr = xr
    for b in ref:
        yr = b._body(yr)
    yr.backward(go.double())
    pr = dict(ref.named_parameters())
    for k, p in layer.named_parameters():
        assert _err(p.grad - seed[k], pr[k].grad) < 2e-5, (k, _err(p.grad - seed[k], pr[k].grad))
    assert _err(xd.grad, xr.grad) < 2e-5
