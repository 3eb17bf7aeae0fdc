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
