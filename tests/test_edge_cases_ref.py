"""Target-building edge cases of LSHead's segm / pose tasks against the reference run live in the harness (skipped
where /root/reference is absent): a ragged batch (one image smaller than the padded batch: part of every level's grid
is masked out) must give the same losses; an image without instances makes the reference raise in its ground-truth
preparation -- the same error is raised here (same error behaviour)."""
import copy
import os

import numpy as np
import pytest
import torch

from tests import golden_util as gu

pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/code'), reason='the reference tree is not on this machine')


def _heads(task):
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    from oracle.ref_harness import bootstrap
    bootstrap.load_reference()
    import mmcv
    from mmdet.models import build_head as ref_build
    from lsnet_amd.models import build_head
    from lsnet_amd.utils import ConfigDict
    cfg, tr, te = gu.head_cfg(task)
    rc = mmcv.Config(copy.deepcopy(cfg))._cfg_dict
    rc.update(train_cfg=mmcv.Config(tr), test_cfg=mmcv.Config(te))
    mc = ConfigDict(copy.deepcopy(cfg))
    mc.update(train_cfg=ConfigDict(tr), test_cfg=ConfigDict(te))
    return gu.fill_params(ref_build(rc), seed=7).train(), gu.fill_params(build_head(mc), seed=7).train()


def _run(task, heads, boxes, labels, ext, metas):
    masks = [gu.make_polygons(b) for b in boxes]
    kps = [gu.make_keypoints(200 + i, b) if len(b) else torch.zeros(0, 51) for i, b in enumerate(boxes)]
    out = []
    for i, head in enumerate(heads):
        kw = dict(gt_bboxes=[b.clone() for b in boxes], gt_extremes=ext if task == 'pose_bbox' else None,
                  gt_keypoints_vs=[k.clone() for k in kps] if 'pose' in task else None,
                  gt_masks=masks if task == 'segm' else None, gt_labels=labels, img_metas=metas)
        try:
            outs = head([f.clone() for f in gu.head_inputs(11)])
            losses = head.loss(*outs, **kw) if i == 0 else head.loss(
                *outs, kw['gt_bboxes'], kw['gt_extremes'], kw['gt_keypoints_vs'], kw['gt_masks'], labels, metas)
            out.append({k: np.array([float(x) for x in v]) for k, v in losses.items()})
        except Exception as e:                                   # noqa: BLE001 -- the error itself is compared
            out.append(e)
    return out


@pytest.mark.parametrize('task', ['segm', 'pose_bbox', 'pose_kbox'])
def test_ragged_batch_equals_reference(task, cpu_oracle_backend):
    h, w = gu.HEAD_IMG
    small = (384, 400)
    b0, l0, e0 = gu.make_gt(100, 4, h, w, num_classes=8)
    b1, l1, e1 = gu.make_gt(101, 3, *small, num_classes=8)
    metas = [dict(pad_shape=(h, w, 3), img_shape=(h, w, 3), scale_factor=1.0),
             dict(pad_shape=small + (3,), img_shape=small + (3,), scale_factor=1.0)]
    ref, ours = _run(task, _heads(task), [b0, b1], [l0, l1], [e0, e1], metas)
    assert isinstance(ref, dict) and isinstance(ours, dict), (ref, ours)
    assert sorted(ref) == sorted(ours)
    for k in ref:
        assert np.allclose(ref[k], ours[k], rtol=1e-4, atol=1e-6), (k, ref[k], ours[k])


@pytest.mark.parametrize('task', ['segm', 'pose_bbox', 'pose_kbox'])
def test_image_without_instances_raises_as_reference(task, cpu_oracle_backend):
    h, w = gu.HEAD_IMG
    b0, l0, e0 = gu.make_gt(100, 4, h, w, num_classes=8)
    metas = [dict(pad_shape=(h, w, 3), img_shape=(h, w, 3), scale_factor=1.0) for _ in range(2)]
    ref, ours = _run(task, _heads(task), [b0, b0[:0]], [l0, l0[:0]], [e0, e0[:0]], metas)
    assert isinstance(ref, Exception) and isinstance(ours, Exception), (ref, ours)
    assert type(ref) is type(ours) and str(ref)[:40] == str(ours)[:40]
