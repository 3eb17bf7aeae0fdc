"""Ground-truth formats of the LSNet tasks, host side (the first pieces of the real-data
pipeline).  Restated from the behaviour of the reference, each function citing what it reproduces:

* `resample_polygon`  -- LoadAnnotations.uniformsample (mmdet/datasets/pipelines/loading.py:314-376): a closed polygon
  re-sampled to exactly n vertices, edge by edge in proportion to edge length (or thinned by dropping the start vertices
  of the shortest edges), including its rounding fix-ups;
* `polygon_landmarks` -- LoadAnnotations.unify_polygons (:422-441): tiny components dropped (box fallback), dense
  re-sampling to n*spline points, every spline-th point kept, clockwise orientation, start vertex = the one nearest the
  box's top-centre;
* `flip_extremes`, `flip_polygons`, `flip_keypoints` -- mmdet/core/bbox/transforms.py:30-87.

Orientation uses the signed shoelace area instead of shapely's `exterior.is_ccw` (same predicate)."""
import numpy as np
import torch

COCO_FLIP_PAIRS = ((1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))


def resample_polygon(pts, n):
    """pts (P, 2) float -> (n, 2)."""
    pts = np.asarray(pts)
    p = pts.shape[0]
    assert pts.ndim == 2 and pts.shape[1] == 2
    nxt = pts[(np.arange(p) + 1) % p]
    length = np.sqrt(((nxt - pts) ** 2).sum(1))
    order = np.argsort(length)
    if p > n:                                   # thin: keep the start vertices of the n longest edges, in order
        return pts[np.sort(order[p - n:])]
    count = np.round(length * n / length.sum()).astype(np.int32)
    count[count == 0] = 1
    total = int(count.sum())
    if total > n:                               # take the surplus from the longest edges, each keeps >= 1 sample
        surplus, i = total - n, -1
        while surplus > 0:
            e = order[i]
            if count[e] > surplus:
                count[e] -= surplus
                surplus = 0
            else:
                surplus -= count[e] - 1
                count[e] = 1
                i -= 1
    elif total < n:                             # give the deficit to the longest edge
        count[order[-1]] += n - total
    assert int(count.sum()) == n
    out = []
    for i in range(p):
        w = (np.arange(count[i], dtype=np.float32) / count[i]).reshape(-1, 1)
        out.append(pts[i:i + 1] * (1 - w) + nxt[i:i + 1] * w)
    return np.concatenate(out, 0)


def signed_area(poly):
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _is_ccw(poly):
    """counter-clockwise in a y-up frame (what shapely's LinearRing.is_ccw reports): positive signed area"""
    return signed_area(poly) > 0


def _component_ok(poly):
    w, h = poly[:, 0].max() - poly[:, 0].min(), poly[:, 1].max() - poly[:, 1].min()
    return w >= 1 and h >= 1 and abs(signed_area(poly)) > 5


def _start_at_top_centre(poly):
    tcx, tcy = (poly[:, 0].min() + poly[:, 0].max()) / 2, poly[:, 1].min()
    i = int(((poly[:, 0] - tcx) ** 2 + (poly[:, 1] - tcy) ** 2).argmin())
    return np.roll(poly, -i, axis=0)


def polygon_landmarks(polygons, gt_bbox, num_points=36, spline_num=10):
    """COCO polygon components of ONE instance -> list of flat (2*num_points,) contours, one per kept component."""
    comps = [np.asarray(p).reshape(-1, 2) for p in polygons]
    comps = [c for c in comps if _component_ok(c)]
    if not comps:
        x1, y1, x2, y2 = (gt_bbox[i] for i in range(4))
        comps = [np.stack([np.stack([x1, y1]), np.stack([x1, y2]), np.stack([x2, y2]), np.stack([x2, y1])])]
    out = []
    for c in comps:
        dense = resample_polygon(c, num_points * spline_num)
        first = int(((dense - dense[0]) ** 2).sum(1).argmin())     # always 0 unless vertices coincide; kept as in the reference
        pts = np.roll(dense, -first, axis=0)[::spline_num]
        if _is_ccw(pts):
            pts = pts[::-1]
        out.append(_start_at_top_centre(pts).reshape(-1))
    return out


# ---------------------------------------------------------------------------------------------------------
def flip_extremes(extremes, img_shape, direction='horizontal'):
    """(..., 8k) blocks [tx,ty, lx,ly, bx,by, rx,ry]: mirror and swap the pair that changes sides."""
    assert direction in ('horizontal', 'vertical')
    out = extremes.clone()
    if direction == 'horizontal':
        w = img_shape[1]
        out[..., 0::8] = w - extremes[..., 0::8]
        out[..., 4::8] = w - extremes[..., 4::8]
        out[..., 2::8], out[..., 3::8] = w - extremes[..., 6::8], extremes[..., 7::8]
        out[..., 6::8], out[..., 7::8] = w - extremes[..., 2::8], extremes[..., 3::8]
    else:
        h = img_shape[0]
        out[..., 3::8] = h - extremes[..., 3::8]
        out[..., 7::8] = h - extremes[..., 7::8]
        out[..., 0::8], out[..., 1::8] = extremes[..., 4::8], h - extremes[..., 5::8]
        out[..., 4::8], out[..., 5::8] = extremes[..., 0::8], h - extremes[..., 1::8]
    return out


def flip_polygons(polygons, img_shape, direction='horizontal'):
    """(N, 2m) contours: mirror, then reverse the vertex order keeping the first vertex first (stays clockwise)."""
    assert direction in ('horizontal', 'vertical')
    out = polygons.clone()
    dim, idx = (img_shape[1], 0) if direction == 'horizontal' else (img_shape[0], 1)
    out[:, idx::2] = dim - out[:, idx::2]
    if out.shape[0] > 0:
        pts = out.reshape(out.shape[0], -1, 2)
        out = torch.cat([pts[:, :1], torch.flip(pts[:, 1:], [1])], 1).reshape(out.shape[0], -1)
    return out


def flip_keypoints(kps, img_shape, direction='horizontal'):
    """(N, 2*17) COCO keypoints: mirror and swap the left/right joints."""
    assert direction in ('horizontal', 'vertical')
    out = kps.clone()
    dim, idx = (img_shape[1], 0) if direction == 'horizontal' else (img_shape[0], 1)
    if out.shape[0] > 0:
        out[:, idx::2] = dim - out[:, idx::2]
        pts = out.reshape(out.shape[0], -1, 2)
        perm = list(range(pts.shape[1]))
        for a, b in COCO_FLIP_PAIRS:
            perm[a], perm[b] = perm[b], perm[a]
        out = pts[:, perm].reshape(out.shape[0], -1)
    return out
