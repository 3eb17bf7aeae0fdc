"""Multi-scale test-time augmentation by instance voting (mmdet/models/detectors/lsnet.py:139-299; the reference
follows ATSS): detections of all scales / flips are mapped back to the original image, then, per class, greedily
clustered by IoU with the current best box; a cluster is replaced by its score-weighted mean (box and landmark
vectors) carrying the cluster's best score, and its members re-enter with score * (1 - IoU) when that is >= 0.05.

Host-side, O(n^2) greedy and sequential like the reference's (it runs once per image on a few thousand boxes)."""
import numpy as np
import torch

from ..data.gt_formats import flip_extremes, flip_keypoints, flip_polygons


def remove_boxes(boxes, min_scale, max_scale):
    """indices of boxes with min_scale^2 <= area <= max_scale^2 (lsnet.py:158-163)"""
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    return torch.nonzero((area >= min_scale * min_scale) & (area <= max_scale * max_scale), as_tuple=False).squeeze(1)


def instances_vote(boxes, vectors, scores, vote_thresh=0.66, keep_thresh=0.05, eps=1e-6):
    """boxes (n,4), vectors (n,2m), scores (n,) tensors -> voted (boxes, vectors, scores), score-descending.
    n <= 1 gives empty results, as in the reference (lsnet.py:229-299)."""
    dev, m2 = boxes.device, vectors.shape[1]
    det = np.concatenate([boxes.detach().cpu().numpy(), scores.detach().cpu().numpy().reshape(-1, 1),
                          vectors.detach().cpu().numpy()], axis=1)
    if det.shape[0] <= 1:
        return boxes.new_zeros((0, 4)), vectors.new_zeros((0, m2)), scores.new_zeros((0,))
    det = det[det[:, 4].argsort()[::-1]]
    out = []
    while det.shape[0] > 0:
        area = (det[:, 2] - det[:, 0]) * (det[:, 3] - det[:, 1])
        w = np.maximum(0.0, np.minimum(det[0, 2], det[:, 2]) - np.maximum(det[0, 0], det[:, 0]))
        h = np.maximum(0.0, np.minimum(det[0, 3], det[:, 3]) - np.maximum(det[0, 1], det[:, 1]))
        inter = w * h
        iou = inter / np.maximum(area[0] + area - inter, eps)
        iou[0] = 1
        idx = np.where(iou >= vote_thresh)[0]
        group, giou = det[idx], iou[idx]
        det = np.delete(det, idx, 0)
        if idx.shape[0] <= 1:
            out.append(group)
            continue
        soft = group.copy()
        soft[:, 4] = soft[:, 4] * (1 - giou)
        soft = soft[soft[:, 4] >= keep_thresh]
        wsum = group[:, 4:5].sum()
        merged = np.zeros((1, 5 + m2))
        merged[0, :4] = (group[:, :4] * group[:, 4:5]).sum(0) / wsum
        merged[0, 5:] = (group[:, 5:] * group[:, 4:5]).sum(0) / wsum
        merged[0, 4] = group[:, 4].max()
        out.append(merged)
        if soft.shape[0] > 0:
            out.append(soft)
    dets = np.concatenate(out, 0)
    dets = dets[dets[:, 4].argsort()[::-1]]
    t = torch.from_numpy(np.ascontiguousarray(dets)).float().to(dev)
    return t[:, :4], t[:, 5:], t[:, 4]


def instance_mapping_back(bboxes, vectors, img_shape, scale_factor, flip, task, flip_direction='horizontal'):
    """detections of one augmented view -> coordinates of the original image (core/bbox/transforms.py:115-137)"""
    if flip:
        b = bboxes.clone()
        if flip_direction == 'horizontal':
            b[:, 0::4], b[:, 2::4] = img_shape[1] - bboxes[:, 2::4], img_shape[1] - bboxes[:, 0::4]
        else:
            b[:, 1::4], b[:, 3::4] = img_shape[0] - bboxes[:, 3::4], img_shape[0] - bboxes[:, 1::4]
        bboxes = b
        fl = {'bbox': flip_extremes, 'segm': flip_polygons}.get(task, flip_keypoints)
        vectors = fl(vectors, img_shape, flip_direction)
    sf = torch.as_tensor(np.asarray(scale_factor, dtype=np.float32), device=bboxes.device)
    bboxes = bboxes.reshape(-1, 4) / sf
    vectors = vectors / sf[:2].repeat(vectors.shape[1] // 2)
    return bboxes, vectors


def vote_merge(aug_bboxes, aug_vectors, aug_labels, img_metas, task, num_classes, num_vectors, max_dets=1000):
    """lsnet.py:139-156 + 326-364: map every view back, vote per class, cap at the `max_dets` best."""
    boxes, vecs = [], []
    for b, v, meta in zip(aug_bboxes, aug_vectors, img_metas):
        meta = meta[0] if isinstance(meta, (list, tuple)) else meta
        bb, vv = instance_mapping_back(b[:, :4], v, meta['img_shape'], meta['scale_factor'], meta['flip'], task,
                                       meta.get('flip_direction', 'horizontal'))
        boxes.append(torch.cat([bb, b[:, 4:5]], 1))
        vecs.append(vv)
    boxes, vecs, labels = torch.cat(boxes), torch.cat(vecs), torch.cat(aug_labels)
    ob, ov, ol = [], [], []
    for j in range(num_classes):
        inds = (labels == j).nonzero(as_tuple=False).squeeze(1)
        bj, vj, sj = instances_vote(boxes[inds, :4].view(-1, 4), vecs[inds], boxes[inds, 4])
        if len(bj) > 0:
            ob.append(torch.cat([bj, sj[:, None]], 1))
            ov.append(vj)
            ol.append(torch.full((bj.shape[0],), j, dtype=torch.int64, device=sj.device))
    if not ob:
        return boxes.new_zeros((0, 5)), boxes.new_zeros((0, num_vectors * 2)), boxes.new_zeros((0,), dtype=torch.long)
    ob, ov, ol = torch.cat(ob), torch.cat(ov), torch.cat(ol)
    if ob.shape[0] > max_dets:
        thr, _ = torch.kthvalue(ob[:, 4].cpu(), ob.shape[0] - max_dets + 1)
        keep = torch.nonzero(ob[:, 4] >= thr.item(), as_tuple=False).squeeze(1)
        ob, ov, ol = ob[keep], ov[keep], ol[keep]
    return ob, ov, ol
