"""The fused bottleneck node (lsnet_amd/ops/resblock.py) on the CPU: its orchestration -- which gradient is gated where,
what rides as the residual of which backward-data launch, the pregate flags ResLayer.forward pairs up, the exact
grad_gamma formula of lsn_conv2d_backward_weight_bn -- run with torch statements of the four kernel primitives and
compared with autograd over the plain modules (the reference's own operator sequence, resnet.py:261-301).  The HIP
primitives themselves are checked on the device (tests/test_resblock_gpu.py)."""
import types

import pytest
import torch
import torch.nn.functional as F

from lsnet_amd.models.backbones import resnet as R
from lsnet_amd.ops import conv as conv_ops
from lsnet_amd.ops import resblock

CL = torch.channels_last


def _scale(bn):
    rstd = torch.rsqrt(bn.running_var + bn.eps)
    return bn.weight.detach() * rstd, rstd


class TorchPrims:
    """What the four C entry points compute, in torch (double precision inputs keep the comparison tight)."""
    Conv2d = conv_ops.Conv2d
    calls = []

    @staticmethod
    def conv_fwd_bn(x, w, bn, stride, pad, dil, relu, residual=None):
        a, _ = _scale(bn)
        y = F.conv2d(x, w.detach() * a[:, None, None, None], bn.bias.detach() - bn.running_mean * a, stride, pad, dil)
        if residual is not None:
            y = y + residual
        return (F.relu(y) if relu else y).contiguous(memory_format=CL)

    @staticmethod
    def relu_gate(gy, y):
        TorchPrims.calls.append('gate')
        return gy * (y > 0)

    @staticmethod
    def dgrad(g, w, in_shape, stride, pad, dil, bn=None, residual=None, gate=None, out=None):
        assert resblock.epilogue_ok(w.shape[2], stride, pad, dil) or (residual is None and gate is None)
        ws = w.detach()
        if bn is not None:
            ws = ws * _scale(bn)[0][:, None, None, None]
        r = torch.nn.grad.conv2d_input(tuple(in_shape), ws, g, stride, pad, dil)
        if residual is not None:
            r = r + residual
        if gate is not None:
            r = r * (gate > 0)
        if out is not None:
            out.copy_(r)
            return out
        return r.contiguous(memory_format=CL)

    @staticmethod
    def wgrad_bn(x, g, w, bn, stride, pad, dil, need=(True, True, True)):
        G = torch.nn.grad.conv2d_weight(x, w.shape, g, stride, pad, dil)
        a, rstd = _scale(bn)
        db = g.sum((0, 2, 3))
        dg = ((w.detach() * G).sum((1, 2, 3)) - bn.running_mean * db) * rstd
        return a[:, None, None, None] * G, dg, db


@pytest.fixture()
def torch_prims(monkeypatch):
    TorchPrims.calls = []
    monkeypatch.setattr(resblock, 'K', TorchPrims)
    real_ok = R.fused_block_ok

    def ok_on_cpu(blk, x, numel=None):   # the device condition aside, the product's own conditions
        return type(blk) is R.Bottleneck and x.dim() == 4 and resblock.bottleneck_ok(blk)
    monkeypatch.setattr(R, 'fused_block_ok', ok_on_cpu)
    yield TorchPrims
    assert R.fused_block_ok is ok_on_cpu or real_ok


def _make_layer(inplanes, planes, n, stride, style='pytorch', seed=0):
    torch.manual_seed(seed)
    layer = R.ResLayer(R.Bottleneck, inplanes, planes, n, stride=stride, style=style).double()
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(-1.0, 1.5)
            m.bias.data.normal_(0, 0.3)
            m.running_mean.normal_(0, 0.5)
            m.running_var.uniform_(0.5, 2.0)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=CL)
    # the cases the folded backward of round 3 could not do: gamma exactly zero and tiny
    last = layer[-1]
    last.norm3.weight.data[:5] = 0.0
    last.norm3.weight.data[5:9] = 1e-6
    layer.train()
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    return layer


def _run(layer, x, fused):
    for p in layer.parameters():
        p.grad = None
    x = x.clone().requires_grad_(True)
    if fused:
        y = layer(x)
    else:
        y = x
        for b in layer:
            y = b._body(y)   # the plain operator sequence
    gy = torch.randn(y.shape, dtype=y.dtype, generator=torch.Generator().manual_seed(5))
    y.backward(gy)
    return y.detach(), x.grad, {n: p.grad.clone() for n, p in layer.named_parameters()}


@pytest.mark.parametrize('inplanes,planes,n,stride,style', [
    (16, 8, 3, 2, 'pytorch'),    # projection shortcut, strided 3x3 conv2 (gate in a strided backward-data launch)
    (32, 8, 2, 1, 'pytorch'),    # identity shortcuts only
    (16, 8, 2, 2, 'caffe'),      # strided 1x1 conv1: classes without a tap -> the unfused tail
])
def test_fused_bottleneck_equals_autograd(torch_prims, inplanes, planes, n, stride, style):
    layer = _make_layer(inplanes, planes, n, stride, style)
    x = torch.randn(2, inplanes, 12, 10, dtype=torch.float64).contiguous(memory_format=CL)
    y0, gx0, gp0 = _run(layer, x, fused=False)
    y1, gx1, gp1 = _run(layer, x, fused=True)
    assert torch.allclose(y0, y1, rtol=1e-10, atol=1e-10)
    assert torch.allclose(gx0, gx1, rtol=1e-9, atol=1e-10)
    assert set(gp0) == set(gp1)
    for k in gp0:
        assert torch.allclose(gp0[k], gp1[k], rtol=1e-9, atol=1e-9), k
    # one stand-alone gate pass for the stage's last block; the caffe-style conv1 (1x1 stride 2) cannot carry its
    # epilogue and gates on its own when a fused block precedes it
    assert torch_prims.calls.count('gate') == (1 if style == 'pytorch' else 1)


def test_zero_gamma_gets_its_gradient(torch_prims):
    """zero_init_residual (resnet.py:607-612 sets norm3.weight = 0): grad_gamma = sum dz x_hat is NOT zero there."""
    layer = _make_layer(32, 8, 1, 1)
    x = torch.randn(2, 32, 9, 7, dtype=torch.float64).contiguous(memory_format=CL)
    _, _, gp = _run(layer, x, fused=True)
    g = gp['0.bn3.weight'] if '0.bn3.weight' in gp else gp['0.norm3.weight']
    assert (g[:5].abs() > 1e-6).all()


def test_frozen_or_foreign_blocks_keep_the_modular_path(torch_prims):
    layer = _make_layer(32, 8, 2, 1)
    for p in layer[0].parameters():
        p.requires_grad_(False)
    assert not resblock.bottleneck_ok(layer[0]) and resblock.bottleneck_ok(layer[1])
    x = torch.randn(1, 32, 6, 6, dtype=torch.float64).contiguous(memory_format=CL)
    y0 = x
    for b in layer:
        y0 = b._body(y0)
    assert torch.allclose(layer(x), y0, rtol=1e-10, atol=1e-10)


def test_every_block_is_judged_on_its_own_input(monkeypatch, torch_prims):
    """ResLayer.forward decided for all blocks on the STAGE input; blocks 1 .. n-1 see block 0's output (four
    times the elements in a trainable stride-1 stage 1, a quarter behind a stride-2 block 0).  A block whose own input
    breaks the 32-bit bound now runs the modular path, its neighbours are told so, and nothing asserts."""
    seen = []

    def ok(blk, x, numel=None):
        n = x.numel() if numel is None else numel
        seen.append(n)
        return type(blk) is R.Bottleneck and resblock.bottleneck_ok(blk) and n <= 2 * 16 * 12 * 10   # "too large" beyond the stage input
    monkeypatch.setattr(R, 'fused_block_ok', ok)
    layer = _make_layer(16, 8, 3, 1)          # stride 1, 16 -> 32 channels: blocks 1, 2 see twice the stage input
    x = torch.randn(2, 16, 12, 10, dtype=torch.float64).contiguous(memory_format=CL)
    y0, gx0, gp0 = _run(layer, x, fused=False)
    seen.clear()
    y1, gx1, gp1 = _run(layer, x, fused=True)
    assert seen == [x.numel(), 2 * x.numel(), 2 * x.numel()]      # one decision per block, on that block's input
    assert torch.allclose(y0, y1, rtol=1e-10, atol=1e-10) and torch.allclose(gx0, gx1, rtol=1e-9, atol=1e-10)
    for k in gp0:
        assert torch.allclose(gp0[k], gp1[k], rtol=1e-9, atol=1e-9), k
    layer = _make_layer(16, 8, 3, 2)          # stride 2: the later blocks see half the elements
    seen.clear()
    _run(layer, torch.randn(2, 16, 11, 9, dtype=torch.float64).contiguous(memory_format=CL), fused=True)
    assert seen == [2 * 16 * 11 * 9, 2 * 32 * 6 * 5, 2 * 32 * 6 * 5]


def test_epilogue_ok_matches_the_residue_classes():
    assert resblock.epilogue_ok(3, 2, 1, 1) and resblock.epilogue_ok(1, 1, 0, 1) and resblock.epilogue_ok(3, 1, 1, 1)
    assert not resblock.epilogue_ok(1, 2, 0, 1)


class SinkPrims(TorchPrims):
    """TorchPrims whose weight gradients go to the parameters' own .grad (what the gradient sinks of the data-parallel
    wrapper are for the HIP primitives) -- the condition under which the identical blocks of a stage queue their weight
    gradients and block 1's backward launches them together (ops/resblock.py flush_wgrad_queue)."""
    batches = []

    @staticmethod
    def wgrad_bn_deferrable(w, bn):
        return True

    @staticmethod
    def wgrad_bn(x, g, w, bn, stride, pad, dil, need=(True, True, True)):
        for p, gr in zip((w, bn.weight, bn.bias), TorchPrims.wgrad_bn(x, g, w, bn, stride, pad, dil)):
            p.grad = gr.clone() if p.grad is None else p.grad + gr
        return None, None, None

    @staticmethod
    def wgrad_bn_jobs(jobs, stride, pad, dil):
        SinkPrims.batches.append(len(jobs))
        for x, g, w, bn in jobs:
            SinkPrims.wgrad_bn(x, g, w, bn, stride, pad, dil)


def test_identical_blocks_share_their_weight_gradient_launches(monkeypatch, torch_prims):
    SinkPrims.calls, SinkPrims.batches = [], []
    monkeypatch.setattr(resblock, 'K', SinkPrims)
    layer = _make_layer(16, 8, 4, 2)                 # block 0 with the projection + three identical blocks
    x = torch.randn(2, 16, 12, 10, dtype=torch.float64).contiguous(memory_format=CL)
    y0, gx0, gp0 = _run(layer, x, fused=False)
    y1, gx1, gp1 = _run(layer, x, fused=True)
    assert torch.allclose(y0, y1, rtol=1e-10, atol=1e-10) and torch.allclose(gx0, gx1, rtol=1e-9, atol=1e-10)
    for k in gp0:
        assert torch.allclose(gp0[k], gp1[k], rtol=1e-9, atol=1e-9), k
    assert SinkPrims.batches == [4, 3, 3]            # conv3 of all four blocks; conv2, conv1 of the identical blocks 1 .. 3
    # stride 1: block 0's conv2 has the others' geometry too; a two-block stage has no group
    SinkPrims.batches = []
    layer = _make_layer(16, 8, 3, 1)
    _, _, gp0 = _run(layer, x, fused=False)
    _, _, gp1 = _run(layer, x, fused=True)
    assert all(torch.allclose(gp0[k], gp1[k], rtol=1e-9, atol=1e-9) for k in gp0) and SinkPrims.batches == [3, 3, 2]
    SinkPrims.batches = []
    layer = _make_layer(16, 8, 2, 1)
    _, _, gp0 = _run(layer, x, fused=False)
    _, _, gp1 = _run(layer, x, fused=True)
    assert all(torch.allclose(gp0[k], gp1[k], rtol=1e-9, atol=1e-9) for k in gp0) and SinkPrims.batches == []


def test_queued_weight_gradients_survive_a_partial_backward(monkeypatch, torch_prims):
    """ADVICE r4 / VERDICT r4 #12: the queued jobs were launched by block 0's backward only.  torch.autograd.grad w.r.t. the
    LAST block's parameters never runs that node -- the engine callback armed with the first queued job launches the
    queue when the pass ends, and BucketedGradReducer.finish() traps a queue that still holds jobs."""
    SinkPrims.calls, SinkPrims.batches = [], []
    monkeypatch.setattr(resblock, 'K', SinkPrims)
    layer = _make_layer(16, 8, 4, 1)
    x = torch.randn(2, 16, 12, 10, dtype=torch.float64).contiguous(memory_format=CL)   # no gradient w.r.t. x
    for p in layer.parameters():
        p.grad = None
    y = layer(x)
    last = [p for p in layer[3].parameters()]
    # only the last block's backward runs: blocks 0 .. 2 are not on the path to these parameters
    torch.autograd.grad(y, last, torch.ones_like(y), allow_unused=True)
    assert resblock.pending_wgrad_jobs() == 0                  # flushed by the callback, not left behind
    _, _, ref = _run(layer, x, fused=False)
    for n, p in layer[3].named_parameters():
        assert p.grad is not None and torch.allclose(p.grad, ref['3.' + n], rtol=1e-9, atol=1e-9), n
    # the trap itself
    q = resblock.WgQueue()
    q[('k',)] = [(None, None, None, None)]
    resblock._live_queues.add(q)
    from lsnet_amd.parallel.reducer import BucketedGradReducer
    red = BucketedGradReducer([torch.nn.Parameter(torch.zeros(4))])
    with pytest.raises(RuntimeError, match='queued weight-gradient'):
        red.finish()
    q.clear()
    red.finish()
