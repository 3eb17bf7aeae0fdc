"""Multi-class NMS that carries the landmark vectors (extreme points / polygon / keypoints) of each
box through suppression (mmdet/core/post_processing/bbox_nms.py:60-99)."""
import torch

from ..ops.nms import batched_nms


def multiclass_nms_lsvr(multi_bboxes, multi_pts, multi_scores, npts, score_thr, nms_cfg, max_num=-1,
                        score_factors=None):
    """multi_bboxes (n,4) or (n,4*C); multi_pts (n,2*npts); multi_scores (n,C+1) with the background
    column last.  Returns (dets (k,5), pts (k,2*npts), labels (k,))."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(-1, num_classes, 4)
    pts = multi_pts[:, None].expand(-1, num_classes, multi_pts.shape[-1])
    scores = multi_scores[:, :-1]
    valid = scores > score_thr
    bboxes, pts = bboxes[valid], pts[valid]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid]
    labels = valid.nonzero(as_tuple=False)[:, 1]
    if bboxes.numel() == 0:
        return (multi_bboxes.new_zeros((0, 5)), multi_pts.new_zeros((0, npts * 2)),
                multi_bboxes.new_zeros((0,), dtype=torch.long))
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, pts[keep], labels[keep]


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """The box-only form (bbox_nms.py:7-57), used by the corner-point-verification head: (dets (k,5), labels (k,))."""
    dets, _, labels = multiclass_nms_lsvr(multi_bboxes, multi_bboxes.new_zeros((multi_bboxes.shape[0], 0)), multi_scores,
                                          0, score_thr, nms_cfg, max_num, score_factors)
    return dets, labels
