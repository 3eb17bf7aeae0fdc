"""Corner-point-verification variant: losses, the corner heat-map assigner, corner pooling, the
detector end to end on CPU (native ops served by the oracle backend = test infrastructure).  The head itself is pinned
against the reference in tests/test_golden_host.py::test_cpv_head_forward_loss_backward_decode."""
import json
import os

import numpy as np
import pytest
import torch

from lsnet_amd.core import PointGenerator, PointHMAssigner, gaussian_radius
from lsnet_amd.models.losses import GaussianFocalLoss, SEPFocalLoss, SmoothL1Loss
from lsnet_amd.ops.corner_pool import CornerPool
from tests import golden_util as gu

HAVE_REF = os.path.isdir('/root/reference/code')


def _ref():
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    from oracle.ref_harness import bootstrap
    bootstrap.load_reference()


def test_corner_pool_is_a_running_max_towards_the_border():
    x = torch.randn(2, 3, 5, 7, dtype=torch.float64, requires_grad=True)
    for mode, dim, towards_start in (('top', 2, True), ('bottom', 2, False), ('left', 3, True), ('right', 3, False)):
        y = CornerPool(mode)(x)
        n = x.shape[dim]
        for i in range(n):
            sl = slice(i, n) if towards_start else slice(0, i + 1)      # 'top': max over rows i.. ; 'bottom': rows ..i
            want = x.narrow(dim, sl.start, sl.stop - sl.start).max(dim)[0]
            assert torch.equal(y.select(dim, i), want), (mode, i)
    # gradient: every output routes to the arg-max it picked
    y = CornerPool('top')(x)
    y.sum().backward()
    assert x.grad.sum() == y.numel() and (x.grad >= 0).all()


def test_cpv_losses_against_their_definitions():
    g = gu.gen(3)
    p = torch.rand(50, generator=g) * 0.98 + 0.01
    t = torch.rand(50, generator=g)
    t[::7] = 1.0
    w = (torch.rand(50, generator=g) > 0.2).float()
    got = GaussianFocalLoss(alpha=2.0, gamma=4.0, loss_weight=0.25)(p, t, w, avg_factor=6.0)
    pos = t == 1
    want = (-(p + 1e-12).log() * (1 - p) ** 2 * pos + -(1 - p + 1e-12).log() * p ** 2 * (1 - t) ** 4) * w
    assert torch.allclose(got, 0.25 * want.sum() / 6.0, rtol=1e-6)

    x = torch.randn(60, generator=g, requires_grad=True)
    tt = (torch.rand(60, generator=g) > 0.7).float()
    ww = torch.rand(60, generator=g) + 0.1
    loss = SEPFocalLoss(gamma=2.0, alpha=0.25, loss_weight=0.1)(x, tt, ww, avg_factor=(tt > 0).sum())
    s = x.detach().sigmoid()
    pl = (-torch.log(s[tt == 1]) * (1 - s[tt == 1]) ** 2 * ww[tt == 1] * 0.25).sum() / ww[tt == 1].sum()
    nl = (-torch.log(1 - s[tt < 1]) * s[tt < 1] ** 2 * 0.75).sum() / (tt > 0).sum()
    assert torch.allclose(loss, 0.1 * (pl + nl), rtol=1e-5)
    loss.backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
    none = SEPFocalLoss()(x.detach(), torch.zeros(60), ww, avg_factor=torch.tensor(3))      # no positives at all
    assert torch.isfinite(none) and torch.allclose(none, (-torch.log(1 - s) * s ** 2 * 0.75).sum() / 3, rtol=1e-5)

    a, b = torch.randn(9, 2, generator=g), torch.randn(9, 2, generator=g)
    wt = torch.ones(9, 2)
    d = (a - b).abs()
    want = torch.where(d < 1 / 9, 0.5 * d * d * 9, d - 0.5 / 9).sum() / 4.0
    assert torch.allclose(SmoothL1Loss(beta=1 / 9.0)(a, b, wt, avg_factor=4.0), want, rtol=1e-6)


def _points(h, w):
    return torch.cat([PointGenerator().grid_points((-(-h // s), -(-w // s)), s, 'cpu') for s in (8, 16, 32, 64, 128)])


def test_corner_heatmap_targets():
    h, w = 256, 384
    pts = _points(h, w)
    boxes = torch.tensor([[40., 50., 200., 180.], [41., 51., 120., 90.], [300., 10., 380., 250.]])
    a = PointHMAssigner(gaussian_bump=True, gaussian_iou=0.7)
    hm_tl, off_tl, hm_br, off_br = a.assign_dense(pts, boxes, strides=(8, 16, 32, 64, 128))
    lvl = torch.log2(pts[:, 2]).int()
    for hm, off, corners in ((hm_tl, off_tl, boxes[:, :2]), (hm_br, off_br, boxes[:, 2:])):
        assert hm.min() >= 0 and hm.max() == 1
        for l in range(3, 8):
            on = lvl == l
            pos = (hm == 1) & on
            assert 1 <= int(pos.sum()) <= 3                      # one per gt and level; gts may share a cell
            # a positive sits within half a cell (+ the half-cell grid shift) of some corner, offset points at it
            p = pts[pos][:, :2]
            d = (p[:, None] - corners[None]).abs().max(-1)[0].min(1)[0]
            assert (d <= 2 ** l).all()
            back = p + off[pos] * 2 ** l
            assert ((back[:, None] - corners[None]).abs().sum(-1).min(1)[0] < 1e-3).all()
        assert ((off != 0).any(1) <= (hm == 1)).all()            # offsets only at positives
        assert ((hm > 0) & (hm < 1)).any()                       # the Gaussian bump around them
    r = gaussian_radius((boxes[:, 3] - boxes[:, 1], boxes[:, 2] - boxes[:, 0]), 0.7)
    assert (r > 0).all() and r[0] > r[1]
    flat = PointHMAssigner(gaussian_bump=False).assign_dense(pts, boxes, strides=(8, 16, 32, 64, 128))
    assert flat[0].dtype == torch.long and set(flat[0].unique().tolist()) == {0, 1}
    assert torch.equal(flat[0] == 1, hm_tl == 1) and torch.equal(flat[1], off_tl)
    empty = a.assign(pts, boxes[:0])
    assert empty[0].sum() == 0 and empty[2].numel() == 0 and empty[3].numel() == pts.shape[0]


@pytest.mark.skipif(not HAVE_REF, reason='the reference tree is not on this machine')
def test_corner_heatmap_assigner_equals_reference():
    _ref()
    from mmdet.core.bbox.assigners.point_hm_assigner import PointHMAssigner as Ref
    pts = _points(384, 512)
    for seed, n in ((1, 5), (2, 9), (3, 1), (4, 30)):           # 30 boxes: many shared cells on the coarse levels
        b, l, _ = gu.make_gt(seed, n, 384, 512, num_classes=8)
        for bump in (True, False):
            want, got = Ref(bump, 0.7).assign(pts, b, l), PointHMAssigner(bump, 0.7).assign(pts, b, l)
            for i, (x, y) in enumerate(zip(want, got)):
                assert x.dtype == y.dtype and x.shape == y.shape, (seed, bump, i)
                if x.dtype == torch.float32:
                    assert torch.allclose(x, y, rtol=1e-6, atol=1e-7), (seed, bump, i)
                else:
                    assert torch.equal(x, y), (seed, bump, i)
    keep = torch.rand(pts.shape[0], generator=gu.gen(0)) < 0.7   # a filtered point set (cells outside the image)
    for x, y in zip(Ref(True, 0.7).assign(pts[keep], b, l), PointHMAssigner(True, 0.7).assign(pts[keep], b, l)):
        assert torch.allclose(x.float(), y.float(), rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------
def _cpv_pipeline(scale):
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    return [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True, with_extreme=True),
            dict(type='Resize', img_scale=scale, keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.5),
            dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32), dict(type='LoadRPDV2Annotations'),
            dict(type='RPDV2FormatBundle'),
            dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_sem_map', 'gt_sem_weights', 'gt_extremes'])]


def test_cpv_detector_trains_and_tests_from_coco_files(tmp_path, cpu_oracle_backend):
    """COCO files -> the CPV pipeline of configs/lsnet/lsnet_bbox_cpv_*.py -> LSCPVDetector (R-50): two optimizer steps,
    then single-view and multi-view (NMS and vote) testing."""
    from lsnet_amd.apis import train_detector
    from lsnet_amd.data import build_dataloader, build_dataset
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.parallel import scatter
    from tests.test_data_pipeline import NORM, _write_images
    ann = _write_images(str(tmp_path))
    ds = build_dataset(dict(type='CocoDataset', ann_file=ann, img_prefix=str(tmp_path), pipeline=_cpv_pipeline((480, 384))))
    loader = build_dataloader(ds, samples_per_gpu=1, workers_per_gpu=0, dist=False, shuffle=True, seed=0)
    torch.manual_seed(3)
    torch.set_num_threads(8)
    model, cfg = build_lsnet('bbox_cpv', 'r50')
    assert type(model).__name__ == 'LSCPVDetector' and type(model.bbox_head).__name__ == 'LSCPVHead'
    cfg.total_epochs, cfg.workflow, cfg.checkpoint_config = 1, [('train', 1)], None
    cfg.log_config = dict(interval=1, hooks=[dict(type='TextLoggerHook')])
    lines = []
    before = model.bbox_head.reppoints_hem_tl_score_out.weight.detach().clone()
    runner = train_detector(model, [loader], cfg, distributed=False, logger=lines.append, channels_last=False)
    assert runner.iter == 3 and not torch.equal(before, model.bbox_head.reppoints_hem_tl_score_out.weight)
    logs = runner.outputs['log_vars']
    for k in ('loss_cls', 'loss_bbox_init', 'loss_bbox_refine', 'loss_heatmap', 'loss_offset', 'loss_sem', 'loss'):
        assert np.isfinite(float(logs[k])) and float(logs[k]) > 0, k

    model.eval()
    model.test_cfg.nms_pre, model.test_cfg.max_per_img, model.test_cfg.score_thr = 30, 20, 0.0
    test_ds = build_dataset(dict(type='CocoDataset', ann_file=ann, img_prefix=str(tmp_path), test_mode=True, pipeline=[
        dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=[(240, 192)], flip=True, transforms=[
            dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), dict(type='Normalize', **NORM),
            dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]))
    item = test_ds[0]
    imgs = [i[None] for i in item['img']]
    metas = [[m.data] for m in item['img_metas']]
    with torch.no_grad():
        single = model(imgs[:1], metas[:1], return_loss=False, rescale=True)
        assert len(single) == 80 and all(r.shape[1] == 5 for r in single) and sum(len(r) for r in single) == 20
        merged = model(imgs, metas, return_loss=False, rescale=True)                      # method 'simple': one NMS
        assert len(merged) == 80 and 0 < sum(len(r) for r in merged) <= 20
        model.test_cfg.method, model.test_cfg.scale_ranges = 'vote', [[0, 10000]]
        voted = model(imgs, metas, return_loss=False, rescale=True)
        assert len(voted) == 80 and all(r.shape[1] == 5 for r in voted)
    w, h = metas[0][0]['ori_shape'][1], metas[0][0]['ori_shape'][0]
    allb = np.concatenate(single)
    assert (allb[:, :4] >= -1e-3).all() and (allb[:, [0, 2]] <= w + 1e-3).all() and (allb[:, [1, 3]] <= h + 1e-3).all()
    assert json.dumps([len(r) for r in voted])
    assert scatter is not None


@pytest.mark.skipif(not HAVE_REF, reason='the reference tree is not on this machine')
@pytest.mark.parametrize('case', ['empty', 'ragged'])
@pytest.mark.parametrize('kind', ['bbox', 'cpv'])
def test_target_edge_cases_equal_reference(kind, case, cpu_oracle_backend):
    """Edge cases of the target builders: one image of the batch has no objects ('empty'); one image is smaller than
    the padded batch, so part of every level's grid lies outside it and is masked out ('ragged').  Both heads against
    the reference's, run live in the harness on the same weights and inputs."""
    import copy
    _ref()
    import mmcv
    from mmdet.models import build_head as ref_build
    from lsnet_amd.models import build_head
    from lsnet_amd.utils import ConfigDict
    cfg, tr, te = gu.cpv_head_cfg() if kind == 'cpv' else gu.head_cfg('bbox')
    rc = mmcv.Config(copy.deepcopy(cfg))._cfg_dict
    rc.update(train_cfg=mmcv.Config(tr), test_cfg=mmcv.Config(te))
    mc = ConfigDict(copy.deepcopy(cfg))
    mc.update(train_cfg=ConfigDict(tr), test_cfg=ConfigDict(te))
    heads = [gu.fill_params(ref_build(rc), seed=7).train(), gu.fill_params(build_head(mc), seed=7).train()]
    h, w = gu.HEAD_IMG
    b0, l0, e0 = gu.make_gt(100, 4, h, w, num_classes=8)
    if case == 'empty':
        boxes, labels, ext = [b0, b0[:0]], [l0, l0[:0]], [e0, e0[:0]]
        small = (h, w)
    else:
        small = (384, 400)                                 # >= 9 cells on the coarsest level, as ATSS needs
        b1, l1, e1 = gu.make_gt(101, 3, *small, num_classes=8)
        boxes, labels, ext = [b0, b1], [l0, l1], [e0, e1]
    metas = [dict(pad_shape=(h, w, 3), img_shape=(h, w, 3), scale_factor=1.0),
             dict(pad_shape=small + (3,), img_shape=small + (3,), scale_factor=1.0)]
    sem, wts = gu.make_sem_maps(boxes, labels, h, w, 8)
    got = []
    for i, head in enumerate(heads):
        outs = head([f.clone() for f in gu.head_inputs(11)])
        if kind == 'cpv':
            losses = head.loss(*outs, boxes, ext, sem, wts, labels, metas)
        elif i == 0:
            losses = head.loss(*outs, gt_bboxes=boxes, gt_extremes=ext, gt_keypoints_vs=None, gt_masks=None,
                               gt_labels=labels, img_metas=metas)
        else:
            losses = head.loss(*outs, boxes, ext, None, None, labels, metas)
        got.append({k: np.array([float(x) for x in (v if isinstance(v, (list, tuple)) else [v])]) for k, v in losses.items()})
    assert sorted(got[0]) == sorted(got[1])
    for k in got[0]:
        assert np.allclose(got[0][k], got[1][k], rtol=1e-4, atol=1e-6), (k, got[0][k], got[1][k])
        assert np.isfinite(got[1][k]).all()


def test_synthetic_cpv_batch_has_the_semantic_maps():
    from lsnet_amd.data import synthetic_batch
    d = synthetic_batch('bbox_cpv', 2, 288, 352, boxes_per_img=3, device='cpu', channels_last=False)
    assert d['gt_sem_map'].shape == d['gt_sem_weights'].shape == (2, 80, 36, 44)
    assert d['gt_sem_map'].sum() > 0 and ((d['gt_sem_weights'] > 0) == (d['gt_sem_map'] > 0)).all()
    assert d['gt_extremes'][0].shape == (3, 10)
