"""Pins the CPU oracle (oracle/lsnet_oracle.c): analytic identities, an independent differentiable
torch restatement, and the reference's own known-answer vectors for NMS."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_py as orc
from tests.torch_dcn_ref import torch_dcn


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def test_zero_offset_unit_mask_is_conv2d():
    torch.manual_seed(0)
    x, w, b = torch.randn(2, 16, 13, 21), torch.randn(24, 16, 3, 3) * 0.1, torch.randn(24)
    off, mask = torch.zeros(2, 18, 13, 21), torch.ones(2, 9, 13, 21)
    out = orc.deform_conv_forward(x, w, b, off, mask, 1, 1, 1)
    assert _rel(out, F.conv2d(x, w, b, padding=1)) < 1e-5
    w4 = torch.randn(24, 4, 3, 3) * 0.1
    off, mask = torch.zeros(2, 18, 7, 11), torch.ones(2, 9, 7, 11)
    out = orc.deform_conv_forward(x, w4, None, off, mask, 2, 1, 1, groups=4, out_hw=(7, 11))
    assert _rel(out, F.conv2d(x, w4, None, stride=2, padding=1, groups=4)) < 1e-5


def test_pyramid_scale_one_is_dcn_v1():
    torch.manual_seed(1)
    x, w = torch.randn(2, 8, 9, 12), torch.randn(8, 8, 3, 3)
    off = torch.rand(2, 18, 9, 12) * 4 - 2
    a = orc.deform_conv_forward(x, w, None, off, None, 1, 1, 1)
    b = orc.deform_conv_forward(x, w, None, off, None, 1, 1, 1, scale_h=1.0, scale_w=1.0, out_hw=(9, 12))
    assert torch.equal(a, b)


CASES = [
    dict(name='v2', mask=True),
    dict(name='v2_s2_g4_dg2', mask=True, stride=2, groups=4, dg=2),
    dict(name='v2_dil2', mask=True, dil=2, pad=2),
    dict(name='v1', mask=False),
    dict(name='pyr_down', mask=False, src=(25, 42), dst=(13, 21)),
    dict(name='pyr_up', mask=False, src=(7, 11), dst=(13, 21)),
]


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_forward_backward_against_torch_restatement(case):
    torch.manual_seed(2)
    B, C, Co = 2, 16, 24
    stride, pad, dil = case.get('stride', 1), case.get('pad', 1), case.get('dil', 1)
    groups, dg = case.get('groups', 1), case.get('dg', 1)
    if 'src' in case:
        (Hs, Ws), (Ho, Wo) = case['src'], case['dst']
        sh, sw = Hs / Ho, Ws / Wo
    else:
        Hs, Ws = 13, 21
        Ho, Wo = orc.out_size(Hs, 3, stride, pad, dil), orc.out_size(Ws, 3, stride, pad, dil)
        sh = sw = 1.0
    x = torch.randn(B, C, Hs, Ws, requires_grad=True)
    w = (torch.randn(Co, C // groups, 3, 3) * 0.1).requires_grad_()
    b = torch.randn(Co, requires_grad=True) if case['mask'] else None
    off = (torch.rand(B, dg * 18, Ho, Wo) * 6 - 3).requires_grad_()   # includes out-of-range samples
    mask = torch.rand(B, dg * 9, Ho, Wo).requires_grad_() if case['mask'] else None
    ref = torch_dcn(x, off, mask, w, b, stride, pad, dil, groups, dg, sh, sw)
    go = torch.randn_like(ref)
    wrt = [t for t in (x, off, mask, w, b) if t is not None]
    grads = torch.autograd.grad(ref, wrt, go)
    out = orc.deform_conv_forward(x, w, b, off, mask, stride, pad, dil, groups, dg, sh, sw, out_hw=(Ho, Wo))
    g = orc.deform_conv_backward(x, w, off, mask, go, stride, pad, dil, groups, dg, sh, sw)
    assert _rel(out, ref.detach()) < 1e-5
    names = ['gx', 'goff'] + (['gmask'] if case['mask'] else []) + ['gw'] + (['gb'] if case['mask'] else [])
    for n, gt in zip(names, grads):
        assert _rel(g[n], gt) < 1e-5, n


def test_nms_known_answers():
    # data of the reference's tests/test_ops/test_nms.py:18-24
    dets = np.array([[49.1, 32.4, 51.0, 35.9, 0.1], [49.3, 32.9, 51.0, 35.3, 0.05],
                     [35.3, 11.5, 39.9, 14.5, 0.9], [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)
    keep = orc.nms(torch.from_numpy(dets), 0.6)
    assert keep.tolist() == [2, 0]
    # data of the docstring example, mmdet/ops/nms/nms_wrapper.py:25-34
    dets = np.array([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9], [49.2, 31.8, 51.0, 35.4, 0.5],
                     [35.1, 11.5, 39.1, 15.7, 0.5], [35.6, 11.8, 39.3, 14.2, 0.5], [35.3, 11.5, 39.9, 14.5, 0.4],
                     [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)
    assert len(orc.nms(torch.from_numpy(dets), 0.6)) == 3
    assert orc.nms(torch.zeros(0, 5), 0.5).numel() == 0


def test_focal_against_python_formula():
    # independent formula: py_sigmoid_focal_loss (mmdet/models/losses/focal_loss.py:11-42)
    torch.manual_seed(3)
    lg, tg = torch.randn(1000, 80) * 3, torch.randint(0, 81, (1000,))
    onehot = F.one_hot(tg, 81)[:, :80].float()

    def py_focal(x):
        p = x.sigmoid()
        pt = (1 - p) * onehot + p * (1 - onehot)
        return F.binary_cross_entropy_with_logits(x, onehot, reduction='none') * \
            (0.25 * onehot + 0.75 * (1 - onehot)) * pt.pow(2.0)

    assert (orc.sigmoid_focal_loss_forward(lg, tg, 2.0, 0.25) - py_focal(lg)).abs().max() < 1e-5
    x = lg.clone().requires_grad_()
    d = torch.randn(1000, 80)
    gr, = torch.autograd.grad(py_focal(x), x, d)
    assert (orc.sigmoid_focal_loss_backward(lg, tg, d, 2.0, 0.25) - gr).abs().max() < 1e-5


def test_nms_against_reference_build():
    """orc_nms vs the reference's own nms_cpu.cpp compiled into oracle/_ref (oracle/build_ref.py)."""
    from oracle import build_ref
    if build_ref.build() is None:
        pytest.skip('no /root/reference and no prebuilt oracle/_ref/nms_ext.so')
    ref = build_ref.load()
    g = torch.Generator().manual_seed(11)
    for n, thr in ((1, 0.5), (37, 0.3), (500, 0.5), (2000, 0.65)):
        xy = torch.rand(n, 2, generator=g) * 200
        wh = torch.rand(n, 2, generator=g) * 60 + 1
        sc = torch.rand(n, 1, generator=g)
        if n > 30:   # duplicated boxes (score ties are excluded: the reference orders them by an unstable sort)
            xy[20:23], wh[20:23] = xy[20], wh[20]
        dets = torch.cat([xy, xy + wh, sc], 1)
        assert orc.nms(dets, thr).tolist() == ref.nms(dets, thr).tolist(), (n, thr)
