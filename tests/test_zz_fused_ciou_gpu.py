"""Fused cross-IOU kernels (csrc/loss.hip) against the torch formulation on the device: loss rows and gradient.
The row function itself is checked against autograd on the CPU (tests/test_fused_cross_iou.py); this test covers the
launch, the C ABI and the autograd binding."""
import pytest
import torch

from lsnet_amd.models.losses import CrossIOULoss
from tests.test_fused_cross_iou import _case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [0, 1])
def test_fused_rows_and_gradient_equal_torch(seed, monkeypatch):
    dev = torch.device('cuda:0')
    pred, target, active, anchor, gt = (t.to(dev) for t in _case(seed, n=5000))
    weight = (torch.rand(len(pred), 1, device=dev) > 0.3).float().expand(-1, 20).contiguous()
    loss = CrossIOULoss(loss_weight=2.0, loss_type='bbox')
    outs = []
    for fused in ('0', '1'):
        monkeypatch.setenv('LSNET_FUSED_CIOU', fused)
        p = pred.clone().requires_grad_()
        rows = loss(p, target, weight, reduction_override='none', anchor_pts=anchor, bbox_gt=gt, pos_inds=active)
        total = loss(p, target, weight, avg_factor=7.0, anchor_pts=anchor, bbox_gt=gt, pos_inds=active)
        (rows.sum() + total).backward()
        outs.append((rows.detach(), total.detach(), p.grad.clone()))
    (r0, t0, g0), (r1, t1, g1) = outs
    assert torch.allclose(r0, r1, rtol=1e-4, atol=1e-5) and torch.allclose(t0, t1, rtol=1e-4)
    scale = g0.abs().amax(1, keepdim=True) + 1e-6
    assert float(((g0 - g1).abs() / scale).max()) < 1e-3


def test_head_with_fused_stage_equals_reference_fixture(monkeypatch):
    """LSHead (bbox) with the fused regression stage switched on against the same reference fixture as the default path."""
    from tests import golden_cases as gc
    monkeypatch.setenv('LSNET_FUSED_CIOU', '1')
    gc.head_case('bbox', torch.device('cuda:0'), channels_last=True)


@pytest.mark.parametrize('kind,nv', [('polygon', 36), ('keypoint', 17)])
def test_fused_polygon_and_keypoint_rows_equal_torch(kind, nv, monkeypatch):
    """The segm / pose variants (lsn_cross_iou_rows_forward / _backward) through CrossIOULoss on the device: rows, reduced
    loss and gradient against the torch formulation."""
    from tests.test_fused_cross_iou import _gen_case
    dev = torch.device('cuda:0')
    pred, target, active, anchor, gt, vs = (t.to(dev) for t in _gen_case(3, nv, n=4000, zero_pair=kind == 'keypoint'))
    weight = (torch.rand(len(pred), 1, device=dev) > 0.3).float().expand(-1, pred.shape[1]).contiguous()
    loss = CrossIOULoss(loss_weight=1.5, loss_type=kind)
    kw = dict(anchor_pts=anchor, pos_inds=active)
    kw.update(dict(bbox_gt=gt) if kind == 'polygon' else dict(bbox_gt=None, vs=vs))
    outs = []
    for fused in ('0', '1'):
        monkeypatch.setenv('LSNET_FUSED_CIOU', fused)
        p = pred.clone().requires_grad_()
        rows = loss(p, target, weight, reduction_override='none', **kw)
        total = loss(p, target, weight, avg_factor=11.0, **kw)
        (rows.sum() + total).backward()
        outs.append((rows.detach(), total.detach(), p.grad.clone()))
    (r0, t0, g0), (r1, t1, g1) = outs
    assert torch.isfinite(r1).all() and torch.isfinite(g1).all()
    assert torch.allclose(r0, r1, rtol=1e-4, atol=1e-5) and torch.allclose(t0, t1, rtol=1e-4)
    scale = g0.abs().amax(1, keepdim=True) + 1e-6
    assert float(((g0 - g1).abs() / scale).max()) < 1e-3


@pytest.mark.parametrize('task', ['segm', 'pose_bbox'])
def test_segm_and_pose_heads_with_fused_rows_equal_reference_fixture(task, monkeypatch):
    from tests import golden_cases as gc
    monkeypatch.setenv('LSNET_FUSED_CIOU', '1')
    gc.head_case(task, torch.device('cuda:0'), channels_last=True)
