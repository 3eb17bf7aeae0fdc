"""LSCPVDetector (mmdet/models/detectors/lscpvnet.py:9-296): backbone -> neck -> LSCPVHead.  Training takes the stride-8
semantic maps of the data pipeline next to boxes and extreme points; testing returns per-class box arrays; multi-view
testing merges either by one NMS over all views (`test_cfg.method='simple'`) or by box voting (`'vote'`)."""
import numpy as np
import torch

from ...core import multiclass_nms
from ...core.vote import remove_boxes, vote_merge
from ..builder import DETECTORS
from .lsnet import SingleStageDetector


def bbox2result(bboxes, labels, num_classes):
    """(k,5) + (k,) -> list of per-class (n,5) arrays (mmdet/core/bbox/transforms.py:140-157)"""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    b, l = bboxes.cpu().numpy(), labels.cpu().numpy()
    return [b[l == i, :] for i in range(num_classes)]


def bbox_mapping_back(bboxes, img_shape, scale_factor, flip, flip_direction='horizontal'):
    """boxes of an augmented view -> original image (mmdet/core/bbox/transforms.py:93-101)"""
    if flip:
        out = bboxes.clone()
        if flip_direction == 'horizontal':
            out[..., 0::4], out[..., 2::4] = img_shape[1] - bboxes[..., 2::4], img_shape[1] - bboxes[..., 0::4]
        else:
            out[..., 1::4], out[..., 3::4] = img_shape[0] - bboxes[..., 3::4], img_shape[0] - bboxes[..., 1::4]
        bboxes = out
    return bboxes.view(-1, 4) / bboxes.new_tensor(np.asarray(scale_factor, dtype=np.float32))


@DETECTORS.register_module()
class LSCPVDetector(SingleStageDetector):

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_sem_map=None,
                      gt_sem_weights=None, gt_extremes=None):
        return self.bbox_head.forward_train(self.extract_feat(img), img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore,
                                            gt_sem_map, gt_sem_weights, gt_extremes)

    def simple_test(self, img, img_metas, rescale=False, show=False, out_dir=False):
        head = self.bbox_head
        dets = head.get_bboxes(*head(self.extract_feat(img)), img_metas, rescale=rescale)
        return [bbox2result(b, l, head.num_classes) for b, l in dets][0]

    def aug_test(self, imgs, img_metas, rescale=False, show=False, out_dir=False):
        if self.test_cfg.get('method', 'simple') == 'simple':
            return self.aug_test_simple(imgs, img_metas, rescale)
        return self.aug_test_vote(imgs, img_metas, rescale)

    def aug_test_simple(self, imgs, img_metas, rescale=False):
        """all views decoded without NMS, mapped back, ONE multi-class NMS (lscpvnet.py:88-118)"""
        head, cfg = self.bbox_head, self.test_cfg
        boxes, scores = [], []
        for img, meta in zip(imgs, img_metas):
            b, s = head.get_bboxes(*head(self.extract_feat(img)), meta, cfg, False, False)[0]
            m = meta[0]
            boxes.append(bbox_mapping_back(b, m['img_shape'], m['scale_factor'], m['flip'],
                                           m.get('flip_direction', 'horizontal')))
            scores.append(s)
        det_b, det_l = multiclass_nms(torch.cat(boxes), torch.cat(scores), cfg.score_thr, cfg.nms, cfg.max_per_img)
        if not rescale:
            det_b = det_b.clone()
            det_b[:, :4] *= det_b.new_tensor(np.asarray(img_metas[0][0]['scale_factor'], dtype=np.float32))
        return bbox2result(det_b, det_l, head.num_classes)

    def aug_test_vote(self, imgs, img_metas, rescale=False):
        """every view decoded WITH NMS, size-filtered per scale, mapped back, merged per class by box voting, capped
        at the 1000 best (lscpvnet.py:120-283)"""
        head, cfg = self.bbox_head, self.test_cfg
        boxes, labels = [], []
        for i, (img, meta) in enumerate(zip(imgs, img_metas)):
            b, l = head.get_bboxes(*head(self.extract_feat(img)), meta, cfg, False, True)[0]
            keep = remove_boxes(b, cfg.scale_ranges[i // 2][0], cfg.scale_ranges[i // 2][1])
            boxes.append(b[keep])
            labels.append(l[keep])
        no_vec = [b.new_zeros((b.shape[0], 0)) for b in boxes]
        det_b, _, det_l = vote_merge(boxes, no_vec, labels, img_metas, 'bbox', head.num_classes, 0)
        if not rescale:
            det_b = det_b.clone()
            det_b[:, :4] *= det_b.new_tensor(np.asarray(img_metas[0][0]['scale_factor'], dtype=np.float32))
        return bbox2result(det_b, det_l, head.num_classes)
