"""Deterministic inputs / parameter fills / output summaries shared by the golden-vector generator
(oracle/ref_harness/make_golden.py, runs against the reference in the build container) and by the
parity tests (which run wherever the fixtures under tests/golden/ are).  Pure torch/numpy on CPU:
the same code yields bit-identical inputs on both sides."""
import types
import zlib

import numpy as np
import torch


import re

_PAIRED_OUT = re.compile(r'pts_\w+_(init|refine)_out\.bias$')
# the head's GroupNorm + ReLU pairs: tower layers `<task>_convs.<i>.bn` (a ConvModule names its norm `bn` whatever its type) and
# the norms behind the pyramid convolutions `<task>_GN`
_HEAD_NORM_BIAS = re.compile(r'(_convs\.\d+\.bn|_GN)\.bias$')
HEAD_NORM_BIAS_SHIFT = 3.5


def gen(seed):
    return torch.Generator().manual_seed(int(seed))


def fill_params(model, seed=0, head_norm_shift=None, pair_gap=6.0):
    """Overwrite every parameter / buffer with values drawn from a generator keyed by its NAME, so
    two differently-constructed models with the same state-dict keys get identical weights.
    head_norm_shift / pair_gap: see the comments at their use; the training-curve fixtures pass (0.0, 1.0) -- round 4's fill --
    because a head whose towers output a mean of 3.5 starts SGD from a classification loss of 2 600 and the first iterations
    are the collapse of that loss (the chaotic trajectory rounds 1 - 3 had), which is not what a curve fixture should pin."""
    if head_norm_shift is None:
        head_norm_shift = HEAD_NORM_BIAS_SHIFT
    with torch.no_grad():
        for name, p in sorted(model.state_dict().items()):
            g = gen(zlib.crc32(name.encode()) + seed)
            if not p.dtype.is_floating_point:
                continue
            if name.endswith('running_var'):
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith('running_mean'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif p.dim() >= 2:
                fan_in = p[0].numel()
                scale = 0.3 if 'conv_offset' in name else 1.0
                p.copy_(torch.randn(p.shape, generator=g) * (scale / fan_in ** 0.5))
            elif name.endswith('weight'):          # norm scales
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:                                   # biases
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
                if _HEAD_NORM_BIAS.search(name):
                    # Round 5.  A head activation is relu(gamma x_hat + beta) with x_hat ~ N(0, 1) per
                    # group.  With beta ~ 0 a fixture holds a few million pre-activations with density 0.4 per unit around
                    # zero: tens of them lie within fp32 rounding of the kink, every kernel combination flips different ones
                    # and each flip moves a 9x9 patch of a feature gradient (the outlier budgets of rounds 3 - 4; the 256-channel
                    # head's same-device check sat at 3.9e-2 of a 5e-2 limit).  With beta = 2.5 the gate still closes on
                    # 0.6 % of the elements -- thousands per tensor, enough to catch a wrong gate -- but the density at the kink
                    # is 23 times lower; the 256-channel head still flipped one (0.55 % of level 0, worst 5.0e-2).  With 3.5: the
                    # gate closes on 0.023 % of the elements -- 60 per tensor of the 32-channel fixtures, 500 of the 256-channel
                    # one: a wrong gate still shows in every gradient -- and the density at the kink is 450 times lower.
                    p.add_(head_norm_shift)
                if _PAIRED_OUT.search(name):
                    # The regression outputs are softplus PAIRS (neg, pos) = channels (2i, 2i + 1) of which the head takes
                    # the larger (lsnet_head.py:323-325).  With symmetric biases a few of the ~200 000 pairs of a fixture
                    # are tied to within fp32 rounding, and any change of a convolution's summation order flips them: the
                    # derivative moves to the other channel and a 9x9 patch of a feature gradient changes (round 3 budgeted
                    # 15 % outliers for that).  A bias gap of 2 with a random sign per pair -- five to ten standard
                    # deviations of the pre-activations -- leaves no pair within 1e-4, so the gradient checks need no
                    # outlier budget.
                    # (round 5: +-6 instead of +-1 -- behind the shifted norm biases below the towers' outputs have mean 3.5 and
                    # the regression pre-activations a standard deviation of ~4; a gap of 2 no longer kept the pairs apart and
                    # the segm head's forward flipped a sign at 0.03 % of its samples)
                    sign = ((torch.rand(p.numel() // 2, generator=g) < 0.5).float() * 2 - 1) * pair_gap
                    p.view(-1, 2).add_(torch.stack([sign, -sign], 1))
    return model


def FEAT_GRAD_STRIDE(level):
    """Sampling stride of the head fixtures' feature gradients: every 13th element of the three large levels, EVERY element
    of the two small ones (6 x 8 and 3 x 4 pixels)."""
    return 13 if level < 3 else 1


def summary(t, stride=13):
    """Compact fingerprint of a tensor: strided sample + sums (enough to catch layout, sign and
    scale errors; 1e-3-level comparison is done on the sample)."""
    t = t.detach().double().cpu().reshape(-1)
    return dict(sample=t[::stride].float().numpy(), sum=np.float64(t.sum().item()),
                abssum=np.float64(t.abs().sum().item()), n=np.int64(t.numel()))


def pack(prefix, t, out, stride=13):
    for k, v in summary(t, stride).items():
        out[f'{prefix}/{k}'] = v


STATS = []   # (prefix, worst relative deviation, share of the sample beyond 1e-3, sample size) of every check() call


def stats_report(kind='grad'):
    """One line over the recorded checks whose prefix starts with `kind` (and clears them): what the tolerances of the
    cross-platform gradient checks are set from."""
    rows = [r for r in STATS if r[0].startswith(kind) or r[0].startswith('g')]
    STATS.clear()
    if not rows:
        return 'no checks recorded'
    worst = max(rows, key=lambda r: r[1])
    share = max(rows, key=lambda r: r[2])
    return (f'{len(rows)} gradient checks: worst deviation {worst[1]:.2e} of the range ({worst[0]}); largest share beyond '
            f'1e-3: {100 * share[2]:.2f}% of {share[3]} samples ({share[0]})')


def check(prefix, t, ref, rtol=1e-3, stride=13, max_outlier_frac=0.0, worst_tol=None):
    """Assert `t` matches the fingerprint stored under `prefix` in the npz `ref`.

    The networks under test are piecewise linear (ReLU) and contain hard selections (max over the
    (neg, pos) halves, top-k, argmax): an activation that is within rounding error of a kink on one
    platform and not on the other flips a 0/1 derivative and changes a small neighbourhood of a
    gradient map by a finite amount.  Such isolated flips are legitimate fp32 behaviour (they happen
    between the reference's own CPU and GPU runs too), so up to `max_outlier_frac` of the sampled
    elements may miss the tolerance; everything else must be within `rtol` of the sample's range and
    the global sum must agree."""
    s = summary(t, stride)
    assert int(s['n']) == int(ref[f'{prefix}/n']), (prefix, s['n'], ref[f'{prefix}/n'])
    want = ref[f'{prefix}/sample']
    scale = max(float(np.abs(want).max()), 1e-12)
    rel = np.abs(s['sample'] - want) / scale
    frac_bad = float((rel > rtol).mean())
    STATS.append((prefix, float(rel.max()), float((rel > 1e-3).mean()), len(want)))
    if max_outlier_frac > 0:   # a budget of k % of a 5-element sample (a bias gradient) must still admit one flip
        max_outlier_frac = max(max_outlier_frac, 1.0 / len(want))
    assert frac_bad <= max_outlier_frac, (f'{prefix}: {100 * frac_bad:.2f}% of the sampled elements are off by more '
                                          f'than {rtol:g} of the range (worst {rel.max():.3e})')
    if worst_tol is not None:   # the outliers a budget admits are kink flips, not garbage: bounded by magnitude
        assert float(rel.max()) <= worst_tol, f'{prefix}: worst sampled element off by {rel.max():.3e} of the range'
    denom = max(float(ref[f'{prefix}/abssum']), 1e-12)
    assert abs(float(s['sum']) - float(ref[f'{prefix}/sum'])) / denom < 10 * rtol, f'{prefix}: sum mismatch'
    return float(np.median(rel))


# ---------------------------------------------------------------------------------------------
# synthetic COCO-shaped ground truth
# ---------------------------------------------------------------------------------------------
def make_gt(seed, num, img_h, img_w, num_classes=80, min_size=16., max_size=None):
    """Boxes with centres uniform over the image and log-uniform sizes, labels, and extreme points
    [top, left, bottom, right, centre] as (x, y) pairs lying on the respective box edge."""
    g = gen(seed)
    max_size = max_size or 0.75 * min(img_h, img_w)
    cx = torch.rand(num, generator=g) * img_w
    cy = torch.rand(num, generator=g) * img_h
    w = torch.exp(torch.rand(num, generator=g) * (np.log(max_size) - np.log(min_size)) + np.log(min_size))
    h = torch.exp(torch.rand(num, generator=g) * (np.log(max_size) - np.log(min_size)) + np.log(min_size))
    x1, x2 = (cx - w / 2).clamp(0, img_w - 2), (cx + w / 2).clamp(2, img_w)
    y1, y2 = (cy - h / 2).clamp(0, img_h - 2), (cy + h / 2).clamp(2, img_h)
    x2, y2 = torch.max(x2, x1 + 2), torch.max(y2, y1 + 2)
    boxes = torch.stack([x1, y1, x2, y2], 1)
    labels = torch.randint(0, num_classes, (num,), generator=g)
    u = torch.rand(num, 4, generator=g)
    tx, ly = x1 + u[:, 0] * (x2 - x1), y1 + u[:, 1] * (y2 - y1)
    bx, ry = x1 + u[:, 2] * (x2 - x1), y1 + u[:, 3] * (y2 - y1)
    extremes = torch.stack([tx, y1, x1, ly, bx, y2, x2, ry, (x1 + x2) / 2, (y1 + y2) / 2], 1)
    return boxes, labels, extremes


def make_polygons(boxes, nv=36):
    """nv-vertex clockwise (image coordinates) ellipse inscribed in each box, starting at the vertex
    nearest the top-centre -- the format the reference's data pipeline emits (loading.py:405-441).
    Returned as an object with `.masks` (per instance: list with one flat (2*nv,) float array) and
    `.areas`, which is all LSHead.process_polygons reads."""
    masks, areas = [], []
    for b in boxes.tolist():
        cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, (b[2] - b[0]) / 2, (b[3] - b[1]) / 2
        ang = -np.pi / 2 + 2 * np.pi * np.arange(nv) / nv      # start at the top, clockwise on screen
        pts = np.stack([cx + rx * np.cos(ang), cy + ry * np.sin(ang)], 1).astype(np.float32)
        masks.append([pts.reshape(-1)])
        areas.append(np.pi * rx * ry)
    return types.SimpleNamespace(masks=masks, areas=np.asarray(areas, dtype=np.float32))


def make_keypoints(seed, boxes, nk=17):
    """(G, nk*3) [x, y, v] with keypoints uniform inside the box and v in {0,1,2} (p = .3,.2,.5)."""
    g = gen(seed)
    G = boxes.shape[0]
    u = torch.rand(G, nk, 2, generator=g)
    x = boxes[:, None, 0] + u[..., 0] * (boxes[:, None, 2] - boxes[:, None, 0])
    y = boxes[:, None, 1] + u[..., 1] * (boxes[:, None, 3] - boxes[:, None, 1])
    r = torch.rand(G, nk, generator=g)
    v = (r > 0.3).float() + (r > 0.5).float()
    v[:, 0] = 2.0   # at least one visible keypoint per instance (kbox needs it)
    return torch.stack([x, y, v], 2).reshape(G, -1)


def head_cfg(task, channels=32, num_classes=8):
    """A reduced LSHead config (same code paths as configs/lsnet/*, small tensors)."""
    norm_cfg = dict(type='GN', num_groups=8 if channels < 256 else 32, requires_grad=True)   # (GN32 at the real width)
    nv = {'bbox': 4, 'segm': 36, 'pose_bbox': 17, 'pose_kbox': 17}[task]
    cfg = dict(type='LSHead', task=task, num_vectors=nv, num_classes=num_classes, in_channels=channels,
               feat_channels=channels, point_feat_channels=channels, stacked_convs=3, num_kernel_points=9,
               gradient_mul=0.1, point_strides=[8, 16, 32, 64, 128], point_base_scale=4, norm_cfg=norm_cfg,
               conv_module_type='dcn',
               loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0))
    if task in ('bbox', 'pose_bbox'):
        cfg.update(loss_bbox_init=dict(type='CrossIOULoss', loss_weight=1.0),
                   loss_bbox_refine=dict(type='CrossIOULoss', loss_weight=2.0))
    if task == 'segm':
        cfg.update(loss_segm_init=dict(type='CrossIOULoss', loss_weight=1.0, loss_type='polygon'),
                   loss_segm_refine=dict(type='CrossIOULoss', loss_weight=2.0, loss_type='polygon'))
    if task in ('pose_bbox', 'pose_kbox'):
        cfg.update(loss_pose_init=dict(type='CrossIOULoss', loss_weight=10.0, loss_type='keypoint'),
                   loss_pose_refine=dict(type='CrossIOULoss', loss_weight=20.0, loss_type='keypoint'))
    train_cfg = dict(init=dict(assigner=dict(type='CentroidAssigner', scale=4, pos_num=1, iou_type='center'),
                               allowed_border=-1, pos_weight=-1, debug=False),
                     refine=dict(assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1, pos_weight=-1,
                                 debug=False))
    test_cfg = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_thr=0.6),
                    max_per_img=100)
    return cfg, train_cfg, test_cfg


# ---------------------------------------------------------------------------------------------
# decode fixture (get_bboxes / _get_bboxes_single -- "bit-exact top-k indices")
# ---------------------------------------------------------------------------------------------
DECODE_CFG = dict(nms_pre=48, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_thr=0.6), max_per_img=100)


def decode_inputs(outs, seed, num_classes=8):
    """Synthetic head outputs of the shapes of `outs` (LSHead.forward's seven lists), derived from a seed alone: both sides
    of the decode fixture -- the reference's get_bboxes in the build container, this package's on whatever device -- decode
    the SAME BITS, so every index result (per-level top-k, label, NMS keep order, the max_per_img cut) must agree exactly.
    What may still differ between two platforms is the last bit of a sigmoid or of a decoded coordinate; the inputs are
    built so that no decision hangs on one:
      * classification: a background of logits near -9 (scores ~1e-4, far below score_thr) and, around a few centres per
        level, candidate (point, class) cells whose SCORES are a random permutation of an evenly spaced grid over
        [0.06, 0.96] -- any two candidates of an image differ by >= 0.9 / N (checked by the generator: > 1e-3), against
        the ~1e-7 two sigmoid implementations may disagree by.  Level 0 gets more candidate points than DECODE_CFG's
        nms_pre, so the per-level top-k cuts real candidates;
      * regression maps: positive values (the softplus range), uniform in [0.05, 2.55].
    Returns (outs', min_gap): the same nested structure with CPU tensors, and the smallest score gap between two candidates
    of one image."""
    g = gen(seed)
    new = [[None if t is None else None for t in lv] for lv in outs]
    for li in range(1, len(outs)):
        for i, t in enumerate(outs[li]):
            if t is not None:
                new[li][i] = torch.rand(tuple(t.shape), generator=g) * 2.5 + 0.05
    cls = outs[0]
    B = cls[0].shape[0]
    maps = [-9.0 - torch.rand(tuple(t.shape), generator=g) for t in cls]
    min_gap = 1.0
    for b in range(B):
        cells = []   # (level, class, y, x)
        for lvl, t in enumerate(cls):
            H, W = t.shape[-2:]
            ncent = max(1, min(16, (H * W) // 48))
            for _ in range(ncent):
                cy, cx = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
                c0 = int(torch.randint(0, num_classes, (1,), generator=g))
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        y, x = cy + dy, cx + dx
                        if not (0 <= y < H and 0 <= x < W):
                            continue
                        c = c0 if float(torch.rand(1, generator=g)) < 0.8 else int(torch.randint(0, num_classes, (1,), generator=g))
                        cells.append((lvl, c, y, x))
        cells = sorted(set(cells))
        n = len(cells)
        perm = torch.randperm(n, generator=g).double()
        score = 0.06 + 0.9 * (perm + 0.5) / n
        logit = torch.log(score / (1 - score)).float()
        for (lvl, c, y, x), v in zip(cells, logit):
            maps[lvl][b, c, y, x] = v
        got = torch.sort(logit.sigmoid().double())[0]
        min_gap = min(min_gap, float((got[1:] - got[:-1]).min()))
    new[0] = maps
    return new, min_gap


def cpv_decode_inputs(outs, seed, num_classes=8):
    """decode_inputs for LSCPVHead.forward's six lists (cls, bbox_init, bbox_refine, hm_score, hm_offset, sem): the corner
    verification of the decode adds two hard selections -- floor(coordinate / stride), exact IEEE arithmetic on identical bits,
    and the arg-max of 2x2 windows of sigmoid(hm_score) -- so the corner heat-map logits are a random permutation of an evenly
    spaced grid over [-4, 4] per image and channel (neighbouring values differ by >= 8 / cells: 2.6e-3 at stride 8, which no two
    sigmoid implementations reorder), the sub-cell offsets uniform in [0, 1)."""
    new, gap = decode_inputs(outs, seed, num_classes)
    g = gen(seed + 1)
    for i, t in enumerate(outs[3]):
        B, Cc, H, W = t.shape
        n = H * W
        m = torch.empty(B, Cc, n)
        for b in range(B):
            for c in range(Cc):
                m[b, c] = (-4.0 + 8.0 * (torch.randperm(n, generator=g).double() + 0.5) / n).float()
        new[3][i] = m.reshape(B, Cc, H, W)
    for i, t in enumerate(outs[4]):
        new[4][i] = torch.rand(tuple(t.shape), generator=g)
    return new, gap


HEAD_IMG = (384, 512)   # -> grids 48x64, 24x32, 12x16, 6x8, 3x4 (>= 9 cells on every level for ATSS)


def head_inputs(seed, channels=32, batch=2):
    g = gen(seed)
    h, w = HEAD_IMG
    return [torch.randn(batch, channels, -(-h // s), -(-w // s), generator=g) for s in (8, 16, 32, 64, 128)]


def cpv_head_cfg(channels=32, num_classes=8):
    """A reduced LSCPVHead config (same code paths as configs/lsnet/lsnet_bbox_cpv_*, small tensors)."""
    norm_cfg = dict(type='GN', num_groups=8, requires_grad=True)
    cfg = dict(type='LSCPVHead', num_classes=num_classes, in_channels=channels, feat_channels=channels,
               point_feat_channels=channels, stacked_convs=3, shared_stacked_convs=1, first_kernel_size=3, kernel_size=1,
               corner_dim=16, num_points=9, gradient_mul=0.1, point_strides=[8, 16, 32, 64, 128], point_base_scale=4,
               norm_cfg=norm_cfg, conv_module_type='dcn',
               loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
               loss_bbox_init=dict(type='CrossIOULoss', loss_weight=1.0),
               loss_bbox_refine=dict(type='CrossIOULoss', loss_weight=2.0),
               loss_heatmap=dict(type='GaussianFocalLoss', alpha=2.0, gamma=4.0, loss_weight=0.25),
               loss_offset=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
               loss_sem=dict(type='SEPFocalLoss', gamma=2.0, alpha=0.25, loss_weight=0.1))
    _, train_cfg, test_cfg = head_cfg('bbox', channels, num_classes)
    train_cfg['heatmap'] = dict(assigner=dict(type='PointHMAssigner', gaussian_bump=True, gaussian_iou=0.7),
                                allowed_border=-1, pos_weight=-1, debug=False)
    return cfg, train_cfg, test_cfg


def make_sem_maps(boxes_list, labels_list, img_h, img_w, num_classes):
    """Stride-8 box-level class maps and 1/area weights, (B, C, H/8, W/8) each -- the inputs LSCPVHead.loss gets from
    the data pipeline (painted from the largest box to the smallest)."""
    h, w = img_h // 8, img_w // 8
    sem = torch.zeros(len(boxes_list), num_classes, h, w)
    wts = torch.zeros(len(boxes_list), num_classes, h, w)
    for i, (boxes, labels) in enumerate(zip(boxes_list, labels_list)):
        area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        for j in torch.argsort(area, descending=True).tolist():
            x1, y1, x2, y2 = (int(v / 8) for v in boxes[j].tolist())
            sem[i, labels[j], y1:y2 + 1, x1:x2 + 1] = 1
            wts[i, labels[j], y1:y2 + 1, x1:x2 + 1] = 1 / float(area[j])
    return sem, wts


def synthetic_eval_case(seed=0, num_images=14):
    """A COCO-format ground truth (3 categories, polygons, crowd regions as run-length masks, keypoints, an image without
    objects, an unlabelled-keypoint person) and detection records of the three kinds (boxes, polygons, keypoints) that
    hit, miss, duplicate and hallucinate objects -- inputs of the evaluation fixtures.  Pure numpy; returns plain
    python structures (json-serialisable)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    cats = [dict(id=1, name='person', supercategory='person'), dict(id=3, name='car', supercategory='vehicle'),
            dict(id=17, name='cat', supercategory='animal')]
    images, anns = [], []
    aid = 1

    def star(n, cx, cy, r):
        ang = np.sort(rng.rand(n)) * 2 * np.pi
        rad = r * (0.6 + 0.4 * rng.rand(n))
        return np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
    for i in range(num_images):
        w, h = int(rng.randint(120, 400)), int(rng.randint(120, 400))
        img_id = 100 - 3 * i if i % 2 else 7 + i          # unsorted, unique ids
        images.append(dict(id=img_id, file_name=f'{img_id:06d}.jpg', width=w, height=h))
        if i == 3:
            continue                                      # an image without annotations
        for _ in range(int(rng.randint(1, 7))):
            r = float(rng.choice([6, 14, 30, 70]))
            cx, cy = float(rng.uniform(r, w - r)), float(rng.uniform(r, h - r))
            pts = star(int(rng.randint(5, 14)), cx, cy, r).clip([0, 0], [w - 1, h - 1]).round(2)
            x1, y1, x2, y2 = pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()
            crowd = int(rng.rand() < 0.12)
            shoelace = 0.5 * abs(np.dot(pts[:, 0], np.roll(pts[:, 1], 1)) - np.dot(pts[:, 1], np.roll(pts[:, 0], 1)))
            kp = np.concatenate([rng.uniform([x1, y1], [x2, y2], (17, 2)), rng.randint(0, 3, (17, 1))], 1).round(1)
            if rng.rand() < 0.15:
                kp[:, 2] = 0
            seg = [pts.reshape(-1).tolist()]
            if crowd:                                     # crowd regions come as uncompressed run-length masks
                a, b = int(x1) * h + int(y1), int((x2 - x1) * h * 0.5)
                seg = dict(size=[h, w], counts=[a, max(b, 1), h * w - a - max(b, 1)])
            anns.append(dict(id=aid, image_id=img_id, category_id=int(rng.choice([1, 1, 3, 17])), iscrowd=crowd,
                             bbox=[float(x1), float(y1), float(x2 - x1), float(y2 - y1)], area=float(shoelace),
                             segmentation=seg, keypoints=kp.reshape(-1).tolist(), num_keypoints=int((kp[:, 2] > 0).sum())))
            aid += 1
    gt = dict(images=images, categories=cats, annotations=anns)
    boxes, polys, kpts = [], [], []
    for a in anns:
        if rng.rand() < 0.15:
            continue                                      # missed object
        for rep in range(1 + int(rng.rand() < 0.3)):      # sometimes detected twice
            x, y, w, h = a['bbox']
            jit = rng.normal(0, 0.08 + 0.15 * rep, 4) * [w, h, w, h]
            bb = [x + jit[0], y + jit[1], max(w + jit[2], 1.0), max(h + jit[3], 1.0)]
            score = float(np.clip(rng.beta(5, 2) - 0.2 * rep, 0.01, 0.999))
            cat = a['category_id'] if rng.rand() > 0.1 else int(rng.choice([1, 3, 17]))
            boxes.append(dict(image_id=a['image_id'], category_id=cat, bbox=[float(v) for v in bb], score=score))
            if isinstance(a['segmentation'], list):
                p = np.array(a['segmentation'][0]).reshape(-1, 2)
            else:
                p = np.array([[x, y], [x + w, y], [x + w, y + h], [x, y + h]])
            p = p + rng.normal(0, 0.06 * (1 + rep), p.shape) * [w, h]
            polys.append(dict(image_id=a['image_id'], category_id=cat, score=score, polygon=p.reshape(-1).round(2).tolist()))
            k = np.array(a['keypoints']).reshape(17, 3).copy()
            k[:, :2] += rng.normal(0, 0.05 * (1 + rep), (17, 2)) * [w, h]
            k[:, 2] = 1
            kpts.append(dict(image_id=a['image_id'], category_id=1, score=score, keypoints=k.reshape(-1).round(2).tolist(),
                             bbox=[float(v) for v in bb]))
    for _ in range(25):                                   # hallucinations
        im = images[int(rng.randint(len(images)))]
        w, h = rng.uniform(5, 80, 2)
        x, y = rng.uniform(0, im['width'] - w), rng.uniform(0, im['height'] - h)
        rec = dict(image_id=im['id'], category_id=int(rng.choice([1, 3, 17])), score=float(rng.uniform(0.01, 0.6)))
        boxes.append(dict(rec, bbox=[float(x), float(y), float(w), float(h)]))
        polys.append(dict(rec, polygon=[float(x), float(y), float(x + w), float(y), float(x + w), float(y + h), float(x), float(y + h)]))
    return gt, boxes, polys, kpts


def res2net_grad_names(bb):
    """Parameters whose gradients the Res2Net fixture pins: in the first (strided) and the last Bottle2neck of layers 2 - 4
    the first and the last of the per-scale 3x3 (deformable) convs with their offset convs, both 1x1 convs and the last
    norm's scale."""
    names = []
    params = dict(bb.named_parameters())
    for li in (2, 3, 4):
        layer = getattr(bb, f'layer{li}')
        for bi in (0, len(layer) - 1):
            for leaf in ('conv1.weight', 'convs.0.weight', 'convs.0.conv_offset.weight', 'convs.2.weight',
                         'convs.2.conv_offset.bias', 'conv3.weight', 'bn3.weight'):
                n = f'layer{li}.{bi}.{leaf}'
                if n in params and params[n].requires_grad:
                    names.append(n)
    return names


def backbone_grad_names(bb):
    """A few parameters of every stage whose gradients the DCN-backbone fixtures pin: the deformable conv2 (weight and
    offset conv) of the first (strided) and the last block of layers 2-4, and the first trainable 1x1 conv."""
    names = []
    params = dict(bb.named_parameters())
    for li in (2, 3, 4):
        layer = getattr(bb, f'layer{li}')
        for bi in (0, len(layer) - 1):
            for leaf in ('conv2.weight', 'conv2.conv_offset.weight', 'conv2.conv_offset.bias', 'conv3.weight'):
                n = f'layer{li}.{bi}.{leaf}'
                if n in params and params[n].requires_grad:
                    names.append(n)
    return names


CURVE0_HW = (384, 512)   # training-curve fixture of the seed-0 init_weights model: 2 images of this size per iteration
