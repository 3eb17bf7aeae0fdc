"""Host logic of lsnet_amd.ops (autograd plumbing, batching, module mirrors) exercised on CPU with
the oracle registered as the 'cpu' backend (test infrastructure)."""
import pytest
import torch
import torch.nn.functional as F

from lsnet_amd import ops
from tests.torch_dcn_ref import torch_dcn


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def test_modulated_autograd_matches_torch_restatement(cpu_oracle_backend):
    torch.manual_seed(0)
    x = torch.randn(2, 8, 9, 11, requires_grad=True)
    w = (torch.randn(12, 8, 3, 3) * 0.1).requires_grad_()
    b = torch.randn(12, requires_grad=True)
    off = (torch.rand(2, 18, 9, 11) * 4 - 2).requires_grad_()
    m = torch.rand(2, 9, 9, 11).requires_grad_()
    out = ops.modulated_deform_conv(x, off, m, w, b, 1, 1, 1, 1, 1)
    ref = torch_dcn(x, off, m, w, b, 1, 1, 1, 1, 1)
    go = torch.randn_like(ref)
    g1 = torch.autograd.grad(out, [x, off, m, w, b], go)
    g2 = torch.autograd.grad(ref, [x, off, m, w, b], go)
    assert _rel(out.detach(), ref.detach()) < 1e-5
    for a, c in zip(g1, g2):
        assert _rel(a, c) < 1e-5


def test_multi_level_equals_per_level(cpu_oracle_backend):
    torch.manual_seed(1)
    conv = ops.ModulatedDeformConvPack(8, 8, 3, 1, 1)
    torch.nn.init.normal_(conv.conv_offset.weight, std=0.1)
    xs = [torch.randn(2, 8, h, w) for h, w in [(12, 16), (6, 8), (3, 4)]]
    multi = conv.forward_multi(xs)
    for x, o in zip(xs, multi):
        assert torch.allclose(conv(x), o, atol=1e-6)
    # gradients flow to shared weights from every level
    loss = sum(o.square().sum() for o in conv.forward_multi(xs))
    loss.backward()
    gw = conv.weight.grad.clone()
    conv.zero_grad()
    sum(conv(x).square().sum() for x in xs).backward()
    assert _rel(gw, conv.weight.grad) < 1e-5


def test_dcn_pack_zero_init_is_half_conv(cpu_oracle_backend):
    # zero-initialised offset conv: offsets 0, mask sigmoid(0)=0.5 (deform_conv.py:521-525)
    torch.manual_seed(2)
    conv = ops.ModulatedDeformConvPack(6, 10, 3, 1, 1)
    x = torch.randn(1, 6, 7, 9)
    ref = 0.5 * F.conv2d(x, conv.weight, None, padding=1) + conv.bias.view(1, -1, 1, 1)
    assert torch.allclose(conv(x), ref, atol=1e-5)


def test_pyramid_module_and_small_input_padding(cpu_oracle_backend):
    torch.manual_seed(3)
    conv = ops.PyramidDeformConv(4, 6, 3, 1, 1)
    src = torch.randn(2, 4, 2, 2)            # smaller than the kernel -> padded (deform_conv.py:614-621)
    off = torch.rand(2, 18, 2, 2)
    out = conv(src, off, 1.0, 1.0)
    assert out.shape == (2, 6, 2, 2)
    src = torch.randn(2, 4, 10, 14, requires_grad=True)
    off = (torch.rand(2, 18, 5, 7) * 2 - 1).requires_grad_()
    out = conv(src, off, 10 / 5, 14 / 7)
    ref = torch_dcn(src, off, None, conv.weight, None, 1, 1, 1, 1, 1, 2.0, 2.0)
    assert _rel(out.detach(), ref.detach()) < 1e-5
    g1 = torch.autograd.grad(out.sum(), [src, off])
    g2 = torch.autograd.grad(ref.sum(), [src, off])
    for a, c in zip(g1, g2):
        assert _rel(a, c) < 1e-5


def test_error_behaviour(cpu_oracle_backend):
    with pytest.raises(ValueError):   # deform_conv.py:29-31
        ops.deform_conv(torch.zeros(4, 5, 5), torch.zeros(1, 18, 5, 5), torch.zeros(4, 4, 3, 3))
    with pytest.raises(RuntimeError):  # offset grid must match the output grid
        ops.deform_conv(torch.zeros(1, 4, 5, 5), torch.zeros(1, 18, 4, 4), torch.zeros(4, 4, 3, 3), padding=1)
    with pytest.raises(AssertionError):  # bias is not supported by DeformConv (deform_conv.py:309)
        ops.DeformConv(4, 4, 3, bias=True)


def test_focal_and_nms_wrappers(cpu_oracle_backend):
    torch.manual_seed(4)
    lg = torch.randn(50, 80, requires_grad=True)
    tg = torch.randint(0, 81, (50,))
    w = torch.rand(50)
    a = (ops.sigmoid_focal_loss(lg, tg, 2.0, 0.25) * w[:, None]).sum()
    b = ops.sigmoid_focal_loss_sum(lg, tg, w, 2.0, 0.25)
    assert torch.allclose(a, b, rtol=1e-5)
    ga, = torch.autograd.grad(a * 0.5, lg)
    gb, = torch.autograd.grad(b * 0.5, lg)
    assert torch.allclose(ga, gb, atol=1e-6)
    boxes = torch.tensor([[0., 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30]])
    scores = torch.tensor([0.9, 0.8, 0.7])
    dets, keep = ops.batched_nms(boxes, scores, torch.tensor([0, 0, 0]), dict(type='nms', iou_thr=0.5))
    assert keep.tolist() == [0, 2] and dets.shape == (2, 5)
    dets, keep = ops.batched_nms(boxes, scores, torch.tensor([0, 1, 0]), dict(type='nms', iou_thr=0.5))
    assert keep.tolist() == [0, 1, 2]


# ---------------------------------------------------------------------------------------------
# host-side dispatch of the fused / own-kernel modules: on CPU tensors they are the reference's ATen ops
def test_conv_bn_gn_modules_are_aten_on_cpu():
    import torch.nn.functional as F
    from lsnet_amd.ops.batch_norm import bn_act
    from lsnet_amd.ops.conv import Conv2d, hip_conv_ok
    from lsnet_amd.ops.group_norm import GroupNorm
    torch.manual_seed(0)
    x = torch.randn(2, 32, 9, 11)
    conv = Conv2d(32, 48, 3, padding=1)
    assert not hip_conv_ok(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
    assert torch.equal(conv(x), F.conv2d(x, conv.weight, conv.bias, 1, 1))
    assert set(conv.state_dict()) == {'weight', 'bias'}            # checkpoint keys of nn.Conv2d
    gn = GroupNorm(8, 32)
    y = gn.forward_multi([x, x[:, :, :4]], relu=True)
    assert torch.allclose(y[0], F.relu(F.group_norm(x, 8, gn.weight, gn.bias, gn.eps)))
    assert y[1].shape == (2, 32, 4, 11)
    bn = torch.nn.BatchNorm2d(32).eval()
    r = torch.randn_like(x)
    assert torch.allclose(bn_act(bn, x, relu=True, residual=r), F.relu(bn(x) + r))


def test_device_only_dispatch_predicates_are_false_on_cpu():
    """Grouped convolution, the fused cross-IOU rows and the frozen-BN kernel are device paths: on CPU tensors the modules
    keep ATen / the torch formulation (the reference's own operators), and the shape rules of the kernels are what the
    header documents."""
    import torch.nn.functional as F
    from lsnet_amd.models.losses import CrossIOULoss
    from lsnet_amd.ops import cross_iou as fused
    from lsnet_amd.ops.batch_norm import _hip_ok
    from lsnet_amd.ops.conv import Conv2d, hip_group_conv_ok
    torch.manual_seed(3)
    x = torch.randn(1, 256, 9, 11).contiguous(memory_format=torch.channels_last)
    conv = Conv2d(256, 256, 3, padding=1, groups=64, bias=False)
    assert not hip_group_conv_ok(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
    assert torch.equal(conv(x), F.conv2d(x, conv.weight, None, 1, 1, 1, 64))
    pred, target = torch.rand(6, 148) + 0.1, torch.rand(6, 148)
    assert not fused.rows_usable(pred, target, 'polygon') and not fused.usable(pred[:, :20], target[:, :20], 'bbox')
    active = torch.zeros(6, 148, dtype=torch.bool)
    active[:, 0::2] = True
    loss = CrossIOULoss(loss_type='polygon')(pred, target, anchor_pts=torch.rand(6, 2), bbox_gt=torch.rand(6, 4) + 1,
                                             pos_inds=active)
    assert torch.isfinite(loss)
    bn = torch.nn.BatchNorm2d(2048).eval()
    assert not _hip_ok(bn, torch.randn(1, 2048, 3, 3).contiguous(memory_format=torch.channels_last), None)   # CPU tensor


def test_resnet_block_equals_reference_sequence():
    """Bottleneck with the fused bn_act calls == conv/bn/relu/add written out (CPU: ATen both ways)."""
    import torch.nn.functional as F
    from lsnet_amd.models.backbones.resnet import Bottleneck
    torch.manual_seed(1)
    ds = torch.nn.Sequential(torch.nn.Conv2d(16, 32, 1, stride=2, bias=False), torch.nn.BatchNorm2d(32))
    blk = Bottleneck(16, 8, stride=2, downsample=ds).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2)
    x = torch.randn(2, 16, 12, 10)
    out = F.relu(blk.norm1(blk.conv1(x)))
    out = F.relu(blk.norm2(blk.conv2(out)))
    want = F.relu(blk.norm3(blk.conv3(out)) + ds(x))
    assert torch.allclose(blk(x), want, atol=1e-6)
