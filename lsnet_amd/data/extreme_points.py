"""Extreme-point annotations for COCO instances: the `extreme_points` field (4 points + box centre = 10 numbers) that
`CocoDataset` reads and the bbox task regresses.  Behaviour of the reference's generator tools/gen_coco_lsvr.py:16-75
(after ExtremeNet): along the contour, every vertex within 2 % of the box height (width) of the top-/bottom-most
(left-/right-most) vertex counts as extreme, and the extreme point is the midpoint of their span."""
import json

import numpy as np

from .coco_index import CocoIndex

THRESH = 0.02


def _run_around(vals, idx, tol, sign):
    """Indices of the contiguous run around `idx` (in the padded, non-cyclic vertex list) whose value stays within
    `tol` of vals[idx]; `sign` +1 for a minimum, -1 for a maximum."""
    picked, n = [idx], len(vals)
    t = idx + 1
    while t < n and sign * (vals[t] - vals[idx]) <= tol:
        picked.append(t)
        t += 1
    t = idx - 1
    while t >= 0 and sign * (vals[t] - vals[idx]) <= tol:
        picked.append(t)
        t -= 1
    return picked


def extreme_points(pts):
    """pts (P, 2) contour vertices (or mask pixels) -> (4, 2) [top, left, bottom, right]."""
    pts = np.asarray(pts)
    left, top = pts[:, 0].min(), pts[:, 1].min()
    right, bottom = pts[:, 0].max(), pts[:, 1].max()
    w, h = right - left + 1, bottom - top + 1
    ring = np.concatenate([pts[-1:], pts, pts[:1]], 0)            # one vertex of wrap-around on either side
    x, y = ring[:, 0], ring[:, 1]

    def mid(vals, idxs):
        return (vals[idxs].max() + vals[idxs].min()) // 2
    t = _run_around(y, int(np.argmin(y)), THRESH * h, +1)
    b = _run_around(y, int(np.argmax(y)), THRESH * h, -1)
    le = _run_around(x, int(np.argmin(x)), THRESH * w, +1)
    r = _run_around(x, int(np.argmax(x)), THRESH * w, -1)
    return np.array([[mid(x, t), top], [left, mid(y, le)], [mid(x, b), bottom], [right, mid(y, r)]])


def annotation_extreme_points(ann, index=None):
    """One COCO annotation -> list of 10 floats.  Polygon segmentations: all components' vertices concatenated;
    run-length (crowd) segmentations: the mask's foreground pixels as integer (x, y)."""
    seg = ann['segmentation']
    if isinstance(seg, list):
        pts = np.array([v for comp in seg for v in comp]).reshape(-1, 2)
    else:
        mask = index.ann_to_mask(ann)
        ys, xs = np.where(mask > 0)
        pts = np.stack([xs, ys], 1).astype(np.int32)
    x, y, w, h = ann['bbox']
    ex = extreme_points(pts).reshape(-1)
    return np.concatenate([ex, [(x + x + w) / 2.0], [(y + y + h) / 2.0]]).tolist()


def add_extreme_points(ann_path, out_path):
    """instances_{split}.json -> instances_lsvr_{split}.json"""
    with open(ann_path) as f:
        data = json.load(f)
    index = CocoIndex(dataset=data)
    for ann in data['annotations']:
        ann['extreme_points'] = annotation_extreme_points(ann, index)
    with open(out_path, 'w') as f:
        json.dump(data, f)
    return len(data['annotations'])
