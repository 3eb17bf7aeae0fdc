"""Point / box -> ground-truth assignment for the two LSNet stages, and the pseudo sampler.

  init stage   : CentroidAssigner(scale=4, pos_num=1)  mmdet/core/bbox/assigners/centroid_assigner.py:26-93
  refine stage : ATSSAssigner(topk=9)                   mmdet/core/bbox/assigners/atss_assigner.py:29-164
  sampler      : PseudoSampler                          mmdet/core/bbox/samplers/pseudo_sampler.py:22-41

Same arithmetic and tie behaviour as the reference (bit-exact gt indices are part of the parity
contract), written with `where`/`scatter` instead of masked assignment so that nothing depends on
host-side shapes."""
import torch

from ..utils.registry import Registry, build_from_cfg

BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
IOU_CALCULATORS = Registry('IoU calculator')


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_iou_calculator(cfg, default_args=None):
    return build_from_cfg(cfg, IOU_CALCULATORS, default_args)


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """IoU / IoF of xyxy boxes without the +1 convention (iou2d_calculator.py:36-130)."""
    assert mode in ('iou', 'iof')
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new_zeros((rows, 1) if is_aligned else (rows, cols))
    b1 = bboxes1 if is_aligned else bboxes1[:, None, :]
    lt = torch.max(b1[..., :2], bboxes2[..., :2])
    rb = torch.min(b1[..., 2:], bboxes2[..., 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    area1 = (bboxes1[:, 2] - bboxes1[:, 0]) * (bboxes1[:, 3] - bboxes1[:, 1])
    if not is_aligned:
        area1 = area1[:, None]
    if mode == 'iou':
        area2 = (bboxes2[:, 2] - bboxes2[:, 0]) * (bboxes2[:, 3] - bboxes2[:, 1])
        union = area1 + area2 - overlap
    else:
        union = area1
    return overlap / union.clamp(min=eps)


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:

    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in (0, 4, 5) and bboxes2.size(-1) in (0, 4, 5)
        return bbox_overlaps(bboxes1[..., :4], bboxes2[..., :4], mode, is_aligned)

    def __repr__(self):
        return self.__class__.__name__ + '()'


class AssignResult:
    """gt_inds: 0 = background, k > 0 = (1-based) index of the assigned gt."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)


def topk_columns(x, k, segments=None, largest=False):
    """(values, indices), both (nseg * k, G): for every row segment (start, n) of the (P, G) matrix `x` what
    `x[start:start + n].topk(k, dim=0, largest=largest)` returns, indices counted from row 0 of `x` -- the per-level
    candidate search of the assigners (atss_assigner.py:103-111, centroid_assigner.py:74).  On the device all segments
    are one launch of lsn_topk_columns (ATen runs one single-block radix select per column and call); elsewhere torch."""
    if segments is None:
        segments = [(0, x.shape[0])]
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and len(segments) <= 8 and x.shape[1] > 0 \
            and all(n >= k for _, n in segments):
        from ..ops.backend import get_backend
        return get_backend(x).topk_columns(x, k, [s for s, _ in segments], [n for _, n in segments], largest)
    vals, idxs = [], []
    for start, n in segments:
        v, i = x[start:start + n].topk(k, dim=0, largest=largest)
        vals.append(v)
        idxs.append(i + start)
    return torch.cat(vals, dim=0), torch.cat(idxs, dim=0)


def _labels_of(gt_inds, gt_labels):
    if gt_labels is None:
        return None
    pos = gt_inds > 0
    return torch.where(pos, gt_labels[(gt_inds - 1).clamp(min=0)], -1)


def _empty_assignment(ref, num_gts, num_preds, gt_labels, with_overlaps):
    gt_inds = ref.new_zeros((num_preds,), dtype=torch.long)
    labels = None if gt_labels is None else ref.new_full((num_preds,), -1, dtype=torch.long)
    return AssignResult(num_gts, gt_inds, ref.new_zeros((num_preds,)) if with_overlaps else None, labels)


@BBOX_ASSIGNERS.register_module()
class CentroidAssigner:
    """Each gt picks, on the FPN level matching its size, its `pos_num` nearest points (distance
    normalised by the gt's width/height); a point claimed by several gts keeps the nearest."""

    def __init__(self, scale=4, pos_num=3, iou_type='center'):
        self.scale, self.pos_num, self.iou_type = scale, pos_num, iou_type

    def assign(self, points, gt_bboxes, gt_extreme_pts, gt_bboxes_ignore=None, gt_labels=None):
        INF = 1e8
        num_gts, num_points = gt_bboxes.shape[0], points.shape[0]
        if num_gts == 0 or num_points == 0:
            return _empty_assignment(points, num_gts, num_points, gt_labels, False)
        xy = points[:, :2]
        lvl = torch.log2(points[:, 2]).int()
        lvl_min, lvl_max = lvl.min(), lvl.max()
        if self.iou_type == 'centroid':
            centers = self.gen_centroid(gt_extreme_pts, num_gts)
        else:
            centers = (gt_bboxes[:, :2] + gt_bboxes[:, 2:]) / 2
        wh = (gt_bboxes[:, 2:] - gt_bboxes[:, :2]).clamp(min=1e-6)
        gt_lvl = ((torch.log2(wh[:, 0] / self.scale) + torch.log2(wh[:, 1] / self.scale)) / 2).int()
        gt_lvl = torch.clamp(gt_lvl, min=lvl_min, max=lvl_max)
        dist = ((xy[:, None, :] - centers[None, :, :]) / wh[None, :, :]).norm(dim=2)
        dist = torch.where(lvl[:, None] != gt_lvl[None, :], INF, dist)
        near_d, near_i = topk_columns(dist, self.pos_num)
        claimed = torch.full_like(dist, INF).scatter_(0, near_i, near_d)
        best_d, best_gt = claimed.min(dim=1)
        gt_inds = torch.where(best_d != INF, best_gt + 1, 0)
        return AssignResult(num_gts, gt_inds, None, labels=_labels_of(gt_inds, gt_labels))

    @staticmethod
    def gen_centroid(pts, num_gts):
        """Intersection of the two lines joining opposite triangle centroids of the extreme-point
        quadrilateral (centroid_assigner.py:95-139)."""
        ext = pts[:, :-2].reshape(pts.shape[0], -1, 2)      # (G, 4, 2)
        ring = torch.cat([ext, ext], dim=1)                  # indices i..i+2 wrap around
        cen = torch.stack([ring[:, i:i + 3].sum(1) / 3.0 for i in range(4)], dim=1)  # (G, 4, 2)
        (x1, y1), (x2, y2) = (cen[:, 0, 0], cen[:, 0, 1]), (cen[:, 2, 0], cen[:, 2, 1])
        (x3, y3), (x4, y4) = (cen[:, 1, 0], cen[:, 1, 1]), (cen[:, 3, 0], cen[:, 3, 1])
        d1, d2 = x1 * y2 - y1 * x2, x3 * y4 - y3 * x4
        den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
        cx = (d1 * (x3 - x4) - d2 * (x1 - x2)) / den
        cy = (d1 * (y3 - y4) - d2 * (y1 - y2)) / den
        return torch.stack([cx, cy], dim=-1)


def gaussian_radius(det_size, min_overlap):
    """CornerNet's radius: the largest corner displacement that still leaves IoU >= min_overlap with the gt box, the
    minimum over its three cases (point_hm_assigner.py:148-166)."""
    height, width = det_size
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 - torch.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 - torch.sqrt(b2 ** 2 - 16 * c2)) / 8
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + torch.sqrt(b3 ** 2 - 4 * a3 * c3)) / (2 * a3)
    return torch.min(torch.stack([r1, r2, r3], dim=1), dim=1)[0]


@BBOX_ASSIGNERS.register_module()
class PointHMAssigner:
    """Corner heat-map targets of the corner-point-verification head (point_hm_assigner.py:8-146).  For the top-left
    and the bottom-right corner of every gt box, on EVERY level: the grid point nearest to the corner is a positive
    (target 1) and carries the sub-cell offset (corner - point) / stride; with `gaussian_bump` every point within the
    gt's Gaussian radius of a corner gets exp(-d^2 / 2 sigma^2), the maximum over gts.

    `assign_dense` is the form the head uses: targets and a positive MASK, no index lists, no host reads when the
    level strides are passed in.  `assign` is the reference's interface on top of it.  Where several gts pick the
    same point the last gt's offset stays, as with the reference's sequential CPU indexing."""

    def __init__(self, gaussian_bump=False, gaussian_iou=0.7):
        self.gaussian_bump, self.gaussian_iou = gaussian_bump, gaussian_iou

    def _corner(self, xy, lvl, levels, corner, radius, sigma, dtype):
        INF = 1e8
        P, G = xy.shape[0], corner.shape[0]
        dist = (xy[:, None, :] - corner[None, :, :]).norm(dim=2)                 # (P, G)
        if self.gaussian_bump:
            g = torch.exp(-torch.pow(dist, 2) / (2 * sigma * sigma)[None, :])
            g = torch.where(dist >= radius[None, :], -INF, g).max(dim=1)[0]
            hm = torch.where(g != -INF, g, torch.zeros_like(g))
        else:
            hm = xy.new_zeros((P,), dtype=dtype)
        offset = xy.new_zeros((P + 1, 2), dtype=torch.float32)                   # row P: sink for empty levels
        winner = torch.full((P + 1,), -1, dtype=torch.long, device=xy.device)
        order = torch.arange(G, device=xy.device)
        for l in levels:
            on = lvl == l
            d = torch.where(on[:, None], dist, float('inf'))
            best_d, best_p = d.min(dim=0)                                        # (G,)
            best_p = torch.where(torch.isfinite(best_d), best_p, torch.full_like(best_p, P))
            winner.zero_().sub_(1)
            winner.scatter_reduce_(0, best_p, order, 'amax')                     # last gt wins a shared point
            keep = winner[best_p] == order
            tgt = torch.where(keep, best_p, torch.full_like(best_p, P))
            val = (corner - xy[best_p.clamp(max=P - 1)]) / float(2 ** int(l))
            offset[tgt] = val
            pos = torch.zeros(P + 1, dtype=torch.bool, device=xy.device)
            pos[best_p] = True
            hm = torch.where(pos[:P], torch.ones_like(hm), hm)
        return hm, offset[:P]

    def assign_dense(self, points, gt_bboxes, strides=None):
        """-> (hm_tl (P,), offset_tl (P, 2), hm_br, offset_br); positives are hm == 1, negatives hm < 1."""
        P, G = points.shape[0], gt_bboxes.shape[0]
        dtype = torch.float32 if self.gaussian_bump else torch.long
        if P == 0 or G == 0:
            z = points.new_zeros
            return z((P,), dtype=dtype), z((P, 2), dtype=torch.float32), z((P,), dtype=dtype), z((P, 2), dtype=torch.float32)
        xy = points[:, :2]
        lvl = torch.log2(points[:, 2]).int()
        if strides is None:
            levels = range(int(lvl.min()), int(lvl.max()) + 1)
        else:
            import math
            levels = [int(math.log2(s)) for s in strides]
        radius = sigma = None
        if self.gaussian_bump:
            w, h = gt_bboxes[:, 2] - gt_bboxes[:, 0], gt_bboxes[:, 3] - gt_bboxes[:, 1]
            radius = gaussian_radius((h, w), self.gaussian_iou)
            sigma = (2 * radius + 1) / 6
        hm_tl, off_tl = self._corner(xy, lvl, levels, gt_bboxes[:, :2], radius, sigma, dtype)
        hm_br, off_br = self._corner(xy, lvl, levels, gt_bboxes[:, 2:], radius, sigma, dtype)
        return hm_tl, off_tl, hm_br, off_br

    def assign(self, points, gt_bboxes, gt_labels=None):
        hm_tl, off_tl, hm_br, off_br = self.assign_dense(points, gt_bboxes)

        def inds(mask):
            return torch.nonzero(mask, as_tuple=False).squeeze(-1)
        return (hm_tl, off_tl, inds(hm_tl == 1), inds(hm_tl < 1), hm_br, off_br, inds(hm_br == 1), inds(hm_br < 1))


@BBOX_ASSIGNERS.register_module()
class ATSSAssigner:
    """Adaptive training sample selection: per level the `topk` boxes whose centres are nearest to
    the gt centre are candidates; IoU threshold = mean + std over a gt's candidates; positives
    must have their centre inside the gt; ties between gts go to the highest IoU."""

    def __init__(self, topk, iou_calculator=dict(type='BboxOverlaps2D')):
        self.topk = topk
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, num_level_bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        INF = 100000000
        bboxes = bboxes[:, :4]
        num_gt, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        if num_gt == 0 or num_bboxes == 0:
            return _empty_assignment(bboxes, num_gt, num_bboxes, gt_labels, True)
        overlaps = self.iou_calculator(bboxes, gt_bboxes)                      # (N, G)
        gt_c = torch.stack(((gt_bboxes[:, 0] + gt_bboxes[:, 2]) / 2.0, (gt_bboxes[:, 1] + gt_bboxes[:, 3]) / 2.0), 1)
        cx, cy = (bboxes[:, 0] + bboxes[:, 2]) / 2.0, (bboxes[:, 1] + bboxes[:, 3]) / 2.0
        box_c = torch.stack((cx, cy), dim=1)
        dist = (box_c[:, None, :] - gt_c[None, :, :]).pow(2).sum(-1).sqrt()

        segments, start = [], 0
        for n in num_level_bboxes:  # k nearest per level and gt
            segments.append((start, n))
            start += n
        _, cand = topk_columns(dist, self.topk, segments)                       # (L*k, G)
        gcol = torch.arange(num_gt, device=bboxes.device)
        cand_iou = overlaps[cand, gcol]
        thr = cand_iou.mean(0) + cand_iou.std(0)
        is_pos = cand_iou >= thr[None, :]
        ccx, ccy = cx[cand], cy[cand]
        inside = torch.stack([ccx - gt_bboxes[:, 0], ccy - gt_bboxes[:, 1], gt_bboxes[:, 2] - ccx,
                              gt_bboxes[:, 3] - ccy], dim=1).min(dim=1)[0] > 0.01
        is_pos = is_pos & inside

        # (G*N) flat table of the positives' IoU, -INF elsewhere; column-major per gt as in the reference
        # (candidate rows of different levels / gts never collide, so a plain index_put is exact and
        # no data-dependent shape -- hence no host sync -- is involved)
        flat = (cand + gcol[None, :] * num_bboxes).view(-1)
        chosen = torch.zeros(num_gt * num_bboxes, dtype=torch.bool, device=bboxes.device)
        chosen[flat] = is_pos.view(-1)
        iou_t = overlaps.t().contiguous().view(-1)
        table = torch.where(chosen, iou_t, -INF)
        max_overlaps, argmax = table.view(num_gt, -1).t().max(dim=1)
        gt_inds = torch.where(max_overlaps != -INF, argmax + 1, 0)
        return AssignResult(num_gt, gt_inds, max_overlaps, labels=_labels_of(gt_inds, gt_labels))


class SamplingResult:

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            self.pos_gt_bboxes = gt_bboxes.view(-1, 4)[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = None if assign_result.labels is None else assign_result.labels[pos_inds]


@BBOX_SAMPLERS.register_module()
class PseudoSampler:
    """No sampling: every assigned point is a positive, every unassigned one a negative."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result, flags)
