"""Single-stage detector shell and LSDetector (mmdet/models/detectors/single_stage.py:9-121,
detectors/lsnet.py:12-100): backbone -> neck -> LSHead; training returns the head's loss dict,
`simple_test` returns per-class lists of boxes and landmark vectors."""
import numpy as np
import torch
import torch.nn as nn

from ...core.vote import remove_boxes, vote_merge
from ..builder import DETECTORS, build_backbone, build_head, build_neck
from .base import BaseDetector


def bbox_extreme2result(bboxes, extremes, labels, num_classes, width=8):
    """(k,5) boxes + (k,w) vectors + (k,) labels -> [per-class boxes], [per-class vectors]
    (mmdet/core/bbox/transforms.py:198-218)."""
    if bboxes.shape[0] == 0:
        return [[np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)],
                [np.zeros((0, width), dtype=np.float32) for _ in range(num_classes)]]
    b, e, l = bboxes.cpu().numpy(), extremes.cpu().numpy(), labels.cpu().numpy()
    return [[b[l == i, :] for i in range(num_classes)], [e[l == i, :] for i in range(num_classes)]]


def bbox_poly2result(bboxes, polygons, labels, num_classes, num_contour_points):
    return bbox_extreme2result(bboxes, polygons, labels, num_classes, num_contour_points * 2)


@DETECTORS.register_module()
class SingleStageDetector(BaseDetector):

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            for m in (self.neck if isinstance(self.neck, nn.Sequential) else [self.neck]):
                m.init_weights()
        self.bbox_head.init_weights()

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def forward_dummy(self, img):
        return self.bbox_head(self.extract_feat(img))


@DETECTORS.register_module()
class LSDetector(SingleStageDetector):

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_masks=None, gt_extremes=None,
                      gt_keypoints=None, gt_bboxes_ignore=None):
        x = self.extract_feat(img)
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_extremes, gt_keypoints, gt_masks,
                                            gt_labels, gt_bboxes_ignore)

    def simple_test(self, img, img_metas, rescale=False, show=False, out_dir=False):
        head = self.bbox_head
        outs = head(self.extract_feat(img))
        dets = head.get_bboxes(*outs, img_metas, rescale=rescale)
        if head.task == 'bbox':
            results = [bbox_extreme2result(b, v, l, head.num_classes) for b, v, l in dets]
        elif head.task == 'segm' or show or out_dir:
            results = [bbox_poly2result(b, v, l, head.num_classes, head.num_vectors) for b, v, l in dets]
        else:
            # pose: drop boxes of area <= 32^2.  Like the reference (lsnet.py:85-98) only the LAST
            # image of the batch is reported; its test loader always uses one image per batch.
            for b, v, l in dets:
                big = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 1024
                b, v, l = b[big], v[big], l[big]
            results = [bbox_poly2result(b, v, l, head.num_classes, head.num_vectors)]
        return results[0]

    def simple_test_batch(self, img, img_metas, rescale=False):
        """Device-resident per-image (boxes, vectors, labels) for a whole batch -- what the
        inference benchmark times (no host conversion)."""
        head = self.bbox_head
        dets = head.get_bboxes(*head(self.extract_feat(img)), img_metas, rescale=rescale)
        if head.task in ('pose_bbox', 'pose_kbox'):
            out = []
            for b, v, l in dets:
                big = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 1024
                out.append((b[big], v[big], l[big]))
            return out
        return dets

    def aug_test_simple(self, imgs, img_metas, rescale=False):
        """`test_cfg.method == 'simple'` (lsnet.py:27-135): every view decoded without NMS, boxes mapped back to the
        original image, ONE multi-class NMS over all of them; like the reference this returns per-class BOXES only
        (the landmark vectors are dropped by the merge)."""
        from ...core import multiclass_nms
        from .lscpv import bbox2result, bbox_mapping_back
        head, cfg = self.bbox_head, self.test_cfg
        boxes, scores = [], []
        for img, meta in zip(imgs, img_metas):
            b, _, s = head.get_bboxes(*head(self.extract_feat(img)), meta, cfg, False, False)[0]
            m = meta[0]
            boxes.append(bbox_mapping_back(b, m['img_shape'], m['scale_factor'], m['flip'],
                                           m.get('flip_direction', 'horizontal')))
            scores.append(s)
        det_b, det_l = multiclass_nms(torch.cat(boxes), torch.cat(scores), cfg.score_thr, cfg.nms, cfg.max_per_img)
        if not rescale:
            det_b = det_b.clone()
            det_b[:, :4] *= det_b.new_tensor(np.asarray(img_metas[0][0]['scale_factor'], dtype=np.float32))
        return bbox2result(det_b, det_l, head.num_classes)

    def aug_test(self, imgs, img_metas, rescale=False, show=False, out_dir=False):
        """Multi-scale / flip testing by instance voting (lsnet.py:301-417, `test_cfg.method == 'vote'`): every view
        is decoded with NMS, filtered to the box sizes its scale is trusted for (`test_cfg.scale_ranges[i // 2]`, views
        come in (plain, flipped) pairs), mapped back to the original image and merged per class by `instances_vote`."""
        cfg = self.test_cfg
        if cfg.get('method', 'simple') != 'vote':
            return self.aug_test_simple(imgs, img_metas, rescale)
        head = self.bbox_head
        boxes, vecs, labels = [], [], []
        for i, (img, meta) in enumerate(zip(imgs, img_metas)):
            b, v, l = head.get_bboxes(*head(self.extract_feat(img)), meta, cfg, False, True)[0]
            keep = remove_boxes(b, cfg.scale_ranges[i // 2][0], cfg.scale_ranges[i // 2][1])
            boxes.append(b[keep])
            vecs.append(v[keep])
            labels.append(l[keep])
        det_b, det_v, det_l = vote_merge(boxes, vecs, labels, img_metas, head.task, head.num_classes, head.num_vectors)
        if not rescale:
            sf = torch.as_tensor(np.asarray(img_metas[0][0]['scale_factor'], dtype=np.float32), device=det_b.device)
            det_b = det_b.clone()
            det_b[:, :4] *= sf
            det_v = det_v * sf[:2].repeat(det_v.shape[1] // 2)
        if head.task == 'bbox':
            return bbox_extreme2result(det_b, det_v, det_l, head.num_classes)
        if head.task in ('pose_bbox', 'pose_kbox') and not (show or out_dir):
            big = (det_b[:, 2] - det_b[:, 0]) * (det_b[:, 3] - det_b[:, 1]) > 1024
            det_b, det_v, det_l = det_b[big], det_v[big], det_l[big]
        return bbox_poly2result(det_b, det_v, det_l, head.num_classes, head.num_vectors)

