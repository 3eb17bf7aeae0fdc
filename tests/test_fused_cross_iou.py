"""The hand-derived row function of the fused cross-IOU kernel (lsnet_amd/csrc/cross_iou_row.h, shared by the device
kernel in csrc/loss.hip) against the torch formulation and its autograd gradient (models/losses/cross_iou_loss.py).
The header is compiled here with g++ into a scratch library: the same source the GPU runs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from lsnet_amd.models.losses.cross_iou_loss import cross_iou_loss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = r'''
#include "cross_iou_row.h"
extern "C" void rows(const float *p, const float *t, const unsigned char *a, const float *anchor, const float *gt, int n,
                     float alpha, float eps, float *loss, float *grad) {
    for (int i = 0; i < n; ++i) {
        CrossIouRow r;
        cross_iou_bbox_row(p + 20 * i, t + 20 * i, a + 20 * i, anchor + 2 * i, gt + 4 * i, alpha, eps, 1, &r);
        loss[i] = r.loss;
        for (int c = 0; c < 20; ++c) grad[20 * i + c] = r.grad[c];
    }
}
extern "C" void stage_rows(const float *p, const float *g, const float *anchor3, const float *gt, const unsigned char *obj,
                           int n, float base, float alpha, float eps, float *loss, float *grad) {
    for (int i = 0; i < n; ++i) {
        CrossIouRow r;
        cross_iou_bbox_stage_row(p + 20 * i, g + 10 * i, anchor3 + 3 * i, gt + 4 * i, obj[i], base, alpha, eps, 1, &r);
        loss[i] = r.loss;
        for (int c = 0; c < 20; ++c) grad[20 * i + c] = r.grad[c];
    }
}
extern "C" void gen_rows(int kind, const float *p, const float *t, const unsigned char *a, const float *anchor, const float *gt,
                         const float *vs, int n, int nv, int sub, float alpha, float eps, float *loss, float *grad) {
    const int M = 4 * (nv + 1);
    for (int i = 0; i < n; ++i)
        loss[i] = kind == 1 ? cross_iou_polygon_row(p + M * i, t + M * i, a + M * i, anchor + 2 * i, gt + 4 * i, nv, sub, alpha, eps,
                                                    grad + M * i)
                            : cross_iou_keypoint_row(p + M * i, t + M * i, a + M * i, vs + nv * i, nv, alpha, eps, grad + M * i);
}
'''


@pytest.fixture(scope='module')
def rowlib(tmp_path_factory):
    d = tmp_path_factory.mktemp('ciou')
    src = d / 'driver.cpp'
    src.write_text(DRIVER)
    so = d / 'ciou.so'
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                           f'-I{os.path.join(ROOT, "lsnet_amd", "csrc")}', str(src), '-o', str(so)])
    return ctypes.CDLL(str(so))


def _case(seed, n=400):
    g = torch.Generator().manual_seed(seed)
    pred = torch.rand(n, 20, generator=g) * 3 + 0.01
    mag = torch.rand(n, 10, generator=g) * 3 + 0.01
    pos = torch.rand(n, 10, generator=g) > 0.5
    target = torch.zeros(n, 20)
    target[:, 0::2] = torch.where(pos, torch.zeros_like(mag), mag)
    target[:, 1::2] = torch.where(pos, mag, torch.zeros_like(mag))
    active = torch.zeros(n, 20, dtype=torch.bool)
    active[:, 0::2], active[:, 1::2] = ~pos, pos
    anchor = torch.rand(n, 2, generator=g) * 10
    c = anchor + torch.randn(n, 2, generator=g)
    wh = torch.rand(n, 2, generator=g) * 4 + 0.2
    gt = torch.cat([c - wh, c + wh], 1)
    target[::17] = 0                                             # rows without an object: zero targets
    pred[5, 3] = target[5, 3] = 1.25                             # an exact tie of max / min
    return pred, target, active, anchor, gt


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_row_function_matches_torch_and_autograd(rowlib, seed):
    pred, target, active, anchor, gt = _case(seed)
    p = pred.clone().requires_grad_()
    want = cross_iou_loss(p, target, None, reduction='none', loss_type='bbox', anchor_pts=anchor, bbox_gt=gt, pos_inds=active)
    up = torch.rand(len(pred), generator=torch.Generator().manual_seed(9))
    (want * up).sum().backward()
    n = len(pred)
    loss, grad = np.zeros(n, np.float32), np.zeros((n, 20), np.float32)
    f32, u8 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_ubyte)
    arrs = [np.ascontiguousarray(x.numpy(), dtype=np.float32) for x in (pred, target, anchor, gt)]
    act = np.ascontiguousarray(active.numpy().astype(np.uint8))
    rowlib.rows(arrs[0].ctypes.data_as(f32), arrs[1].ctypes.data_as(f32), act.ctypes.data_as(u8), arrs[2].ctypes.data_as(f32),
                arrs[3].ctypes.data_as(f32), n, ctypes.c_float(0.2), ctypes.c_float(1e-6), loss.ctypes.data_as(f32),
                grad.ctypes.data_as(f32))
    np.testing.assert_allclose(loss, want.detach().numpy(), rtol=2e-5, atol=2e-6)
    got = grad * up.numpy()[:, None]
    ref = p.grad.numpy()
    scale = np.abs(ref).max(1, keepdims=True) + 1e-6
    assert (np.abs(got - ref) / scale).max() < 2e-4, float((np.abs(got - ref) / scale).max())


@pytest.mark.parametrize('seed', [3, 4])
def test_stage_row_matches_head_formulation(rowlib, seed):
    """prediction in stride units + extreme points + anchors  ->  rows, as LSHead.loss_levels composes them
    (`_gt_reg`, normalisation by 4 * stride, cross_iou_loss) and its autograd gradient w.r.t. the raw prediction."""
    from lsnet_amd.models.dense_heads.ls_head import LSHead
    g = torch.Generator().manual_seed(seed)
    n = 600
    stride = torch.tensor([8., 16., 32., 64., 128.])[torch.randint(0, 5, (n,), generator=g)]
    anchor = torch.cat([torch.floor(torch.rand(n, 2, generator=g) * 20) * stride[:, None], stride[:, None]], 1)
    raw = torch.rand(n, 20, generator=g) * 2 + 0.01
    centre = anchor[:, :2] + torch.randn(n, 2, generator=g) * stride[:, None] * 2
    half = (torch.rand(n, 2, generator=g) * 3 + 0.3) * stride[:, None]
    box = torch.cat([centre - half, centre + half], 1)
    ext = torch.stack([centre[:, 0], box[:, 1], box[:, 0], centre[:, 1], centre[:, 0], box[:, 3], box[:, 2], centre[:, 1],
                       centre[:, 0], centre[:, 1]], 1) + torch.randn(n, 10, generator=g)
    ext[3, 0] = anchor[3, 0]                                     # dx == 0 exactly: the ">= 0" side
    obj = torch.rand(n, generator=g) > 0.25
    weights = obj.float()[:, None].expand(-1, 20)
    ext_in = torch.where(obj[:, None], ext, torch.zeros_like(ext))       # the target builder zeroes rows without object
    box_in = torch.where(obj[:, None], box, torch.zeros_like(box))
    p = raw.clone().requires_grad_()
    norm = 4 * stride[:, None]
    reg, active = LSHead._gt_reg(ext_in, anchor, weights)
    want = cross_iou_loss(p * stride[:, None] / norm, reg / norm, weights.mean(-1), reduction='none', loss_type='bbox',
                          anchor_pts=anchor[:, :2] / norm, bbox_gt=box_in / norm, pos_inds=active)
    up = torch.rand(n, generator=g)
    (want * up).sum().backward()
    loss, grad = np.zeros(n, np.float32), np.zeros((n, 20), np.float32)
    f32, u8 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_ubyte)
    arrs = [np.ascontiguousarray(x.numpy(), dtype=np.float32) for x in (raw, ext_in, anchor, box_in)]
    ob = np.ascontiguousarray(obj.numpy().astype(np.uint8))
    rowlib.stage_rows(arrs[0].ctypes.data_as(f32), arrs[1].ctypes.data_as(f32), arrs[2].ctypes.data_as(f32),
                      arrs[3].ctypes.data_as(f32), ob.ctypes.data_as(u8), n, ctypes.c_float(4.0), ctypes.c_float(0.2),
                      ctypes.c_float(1e-6), loss.ctypes.data_as(f32), grad.ctypes.data_as(f32))
    w = obj.float().numpy()
    np.testing.assert_allclose(loss * w, want.detach().numpy(), rtol=5e-5, atol=5e-6)
    got, ref = grad * (up.numpy() * w)[:, None], p.grad.numpy()
    scale = np.abs(ref).max(1, keepdims=True) + 1e-6
    assert (np.abs(got - ref) / scale).max() < 5e-4, float((np.abs(got - ref) / scale).max())


def _gen_case(seed, nv, n=300, zero_pair=False):
    """Rows of 4 (nv + 1) components as the segm / pose heads build them: per (neg, pos) pair one active half."""
    g = torch.Generator().manual_seed(seed)
    m = 4 * (nv + 1)
    pred = torch.rand(n, m, generator=g) * 3 + 0.01
    mag = torch.rand(n, m // 2, generator=g) * 3 + 0.01
    pos = torch.rand(n, m // 2, generator=g) > 0.5
    target = torch.zeros(n, m)
    target[:, 0::2] = torch.where(pos, torch.zeros_like(mag), mag)
    target[:, 1::2] = torch.where(pos, mag, torch.zeros_like(mag))
    active = torch.zeros(n, m, dtype=torch.bool)
    active[:, 0::2], active[:, 1::2] = ~pos, pos
    anchor = torch.rand(n, 2, generator=g) * 10
    c = anchor + torch.randn(n, 2, generator=g)
    wh = torch.rand(n, 2, generator=g) * 4 + 0.2
    gt = torch.cat([c - wh, c + wh], 1)
    vs = (torch.rand(n, nv, generator=g) > 0.3).float() * 2            # COCO visibility 0 / 2
    target[::17] = 0                                                   # rows without an object
    pred[5, 3] = target[5, 3] = 1.25                                   # an exact tie of max / min
    if zero_pair:                                                      # a landmark below the eps clamp of the keypoint variant
        pred[7, 8:12] = 0.0                                            # (the polygon variant would divide 0 by 0 there, as the
        target[7, 8:12] = 0.0                                          # reference does)
    return pred, target, active, anchor, gt, vs


@pytest.mark.parametrize('kind,nv', [('polygon', 36), ('polygon', 9), ('keypoint', 17), ('keypoint', 4)])
@pytest.mark.parametrize('seed', [0, 1])
def test_polygon_and_keypoint_rows_match_torch_and_autograd(rowlib, kind, nv, seed):
    pred, target, active, anchor, gt, vs = _gen_case(seed + 10 * nv, nv, zero_pair=kind == 'keypoint')
    p = pred.clone().requires_grad_()
    want = cross_iou_loss(p, target, None, reduction='none', loss_type=kind, anchor_pts=anchor, bbox_gt=gt, pos_inds=active,
                          vs=vs, stride=9)
    up = torch.rand(len(pred), generator=torch.Generator().manual_seed(9))
    (want * up).sum().backward()
    n, m = pred.shape
    loss, grad = np.zeros(n, np.float32), np.zeros((n, m), np.float32)
    f32, u8 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_ubyte)
    arrs = [np.ascontiguousarray(x.numpy(), dtype=np.float32) for x in (pred, target, anchor, gt, vs)]
    act = np.ascontiguousarray(active.numpy().astype(np.uint8))
    rowlib.gen_rows(1 if kind == 'polygon' else 2, arrs[0].ctypes.data_as(f32), arrs[1].ctypes.data_as(f32),
                    act.ctypes.data_as(u8), arrs[2].ctypes.data_as(f32), arrs[3].ctypes.data_as(f32),
                    arrs[4].ctypes.data_as(f32), n, nv, 9, ctypes.c_float(0.2), ctypes.c_float(1e-6),
                    loss.ctypes.data_as(f32), grad.ctypes.data_as(f32))
    np.testing.assert_allclose(loss, want.detach().numpy(), rtol=3e-5, atol=3e-6)
    got = grad * up.numpy()[:, None]
    ref = p.grad.numpy()
    scale = np.abs(ref).max(1, keepdims=True) + 1e-6
    assert (np.abs(got - ref) / scale).max() < 3e-4, float((np.abs(got - ref) / scale).max())
