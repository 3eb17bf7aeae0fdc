"""Generates tests/golden/*.npz by running the REFERENCE's own Python (imported from
/root/reference through bootstrap.py, native ops backed by the CPU oracle) on seeded inputs.

Runs only in the build container.  Fixtures are data (inputs are re-derived from seeds by
tests/golden_util.py; outputs are stored as strided samples + sums, integer results in full).

    python oracle/ref_harness/make_golden.py [--only head_bbox,...]
"""
import argparse
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle.ref_harness import bootstrap  # noqa: E402
from tests import golden_util as gu  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def _save(name, data):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **data)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(data)} arrays')


def _ref_head(task, channels=32):
    import mmcv
    from mmdet.models import build_head
    cfg, train_cfg, test_cfg = gu.head_cfg(task, channels)
    cfg = mmcv.Config(copy.deepcopy(cfg))._cfg_dict     # attribute-style dicts, as Config.fromfile gives
    cfg.update(train_cfg=mmcv.Config(train_cfg), test_cfg=mmcv.Config(test_cfg))
    head = build_head(cfg)
    gu.fill_params(head, seed=7)
    return head


def _gt_for(task, batch=2):
    h, w = gu.HEAD_IMG
    boxes, labels, extremes, masks, kps = [], [], [], [], []
    for i in range(batch):
        b, l, e = gu.make_gt(100 + i, 4 + i, h, w, num_classes=8)
        boxes.append(b); labels.append(l); extremes.append(e)
        masks.append(gu.make_polygons(b))
        kps.append(gu.make_keypoints(200 + i, b))
    metas = [dict(pad_shape=(h, w, 3), img_shape=(h, w, 3), scale_factor=1.0) for _ in range(batch)]
    return boxes, labels, extremes, masks, kps, metas


def golden_head(task, channels=32):
    """(8) full LSHead.forward + loss (+ gradients) and get_bboxes, reference code end to end.  channels = 256: the head at
    the width of configs/lsnet/* (GN32, 256-channel towers), fixture head_<task>_256."""
    head = _ref_head(task, channels)
    head.train()
    feats = [f.requires_grad_() for f in gu.head_inputs(11, channels)]
    outs = head(feats)
    data = {}
    names = ['cls', 'bbox_init', 'bbox_refine', 'segm_init', 'segm_refine', 'pose_init', 'pose_refine']
    for n, lv in zip(names, outs):
        for i, t in enumerate(lv):
            if t is not None:
                gu.pack(f'out/{n}/{i}', t, data)
    boxes, labels, extremes, masks, kps, metas = _gt_for(task)
    kw = dict(gt_bboxes=boxes, gt_labels=labels, img_metas=metas,
              gt_extremes=extremes if task in ('bbox', 'pose_bbox') else None,
              gt_keypoints_vs=[k.clone() for k in kps] if 'pose' in task else None,
              gt_masks=masks if task == 'segm' else None)
    losses = head.loss(*outs, **kw)
    total = 0
    for k, v in losses.items():
        data[f'loss/{k}'] = np.array([float(x) for x in v], dtype=np.float64)
        total = total + sum(v)
    total.backward()
    for i, f in enumerate(feats):
        # the two smallest levels are stored WHOLE (a strided sample cannot see an error confined to one residue class
        # of channels; gu.FEAT_GRAD_STRIDE is shared with the checking side)
        gu.pack(f'grad/feat/{i}', f.grad, data, stride=gu.FEAT_GRAD_STRIDE(i))
    for name, p in sorted(head.named_parameters()):
        if p.grad is not None:
            gu.pack(f'grad/param/{name}', p.grad, data, stride=7)
    # inference: decode + NMS
    head.eval()
    with torch.no_grad():
        dets = head.get_bboxes(*[[None if t is None else t.detach() for t in lv] for lv in outs], metas)
    for i, (b, v, l) in enumerate(dets):
        data[f'det/{i}/bboxes'] = b.numpy()
        data[f'det/{i}/vectors'] = v.numpy()
        data[f'det/{i}/labels'] = l.numpy()
    _save(f'head_{task}' + ('' if channels == 32 else f'_{channels}'), data)


def golden_decode():
    """(a19) the REFERENCE's get_bboxes / _get_bboxes_single (lsnet_head.py:1439-1668: per-level top-k, decode, clamp,
    multiclass NMS, max_per_img) on gu.decode_inputs -- head outputs derived from a seed alone, so the checking side decodes
    the same bits and every index result is held EXACTLY on every platform (tests/golden_cases.py decode_case)."""
    import mmcv
    data = {}
    for task in ('bbox', 'segm', 'pose_bbox', 'pose_kbox'):
        head = _ref_head(task)
        head.eval()
        with torch.no_grad():
            outs = head(gu.head_inputs(11))
            syn, gap = gu.decode_inputs(outs, 77)
            assert gap > 1e-3, (task, gap)
            _, _, _, _, _, metas = _gt_for(task)
            dets = head.get_bboxes(*syn, metas, cfg=mmcv.Config(gu.DECODE_CFG))
        data[f'{task}/min_gap'] = np.float64(gap)
        for i, (b, v, l) in enumerate(dets):
            print(task, 'image', i, len(l), 'detections; top score', float(b[0, 4]) if len(l) else None)
            assert 20 < len(l) <= gu.DECODE_CFG['max_per_img']
            data[f'{task}/{i}/bboxes'] = b.numpy()
            data[f'{task}/{i}/vectors'] = v.numpy()
            data[f'{task}/{i}/labels'] = l.numpy()
    _save('decode', data)


def golden_head_cpv():
    """(f-4) LSCPVHead.forward + loss (+ gradients) and get_bboxes, reference code end to end
    (mmdet/models/dense_heads/lscpvnet_head.py)."""
    import mmcv
    from mmdet.models import build_head
    cfg, train_cfg, test_cfg = gu.cpv_head_cfg()
    cfg = mmcv.Config(copy.deepcopy(cfg))._cfg_dict
    cfg.update(train_cfg=mmcv.Config(train_cfg), test_cfg=mmcv.Config(test_cfg))
    head = build_head(cfg)
    gu.fill_params(head, seed=7)
    head.train()
    feats = [f.requires_grad_() for f in gu.head_inputs(11)]
    outs = head(feats)
    data = {}
    names = ['cls', 'bbox_init', 'bbox_refine', 'hm_score', 'hm_offset', 'sem']
    for n, lv in zip(names, outs):
        for i, t in enumerate(lv):
            gu.pack(f'out/{n}/{i}', t, data)
    boxes, labels, extremes, _, _, metas = _gt_for('bbox')
    sem, wts = gu.make_sem_maps(boxes, labels, *gu.HEAD_IMG, 8)
    losses = head.loss(*outs, boxes, extremes, sem, wts, labels, metas)
    total = 0
    for k, v in losses.items():
        v = v if isinstance(v, (list, tuple)) else [v]
        data[f'loss/{k}'] = np.array([float(x) for x in v], dtype=np.float64)
        total = total + sum(v)
    total.backward()
    for i, f in enumerate(feats):
        gu.pack(f'grad/feat/{i}', f.grad, data)
    for name, p in sorted(head.named_parameters()):
        if p.grad is not None:
            gu.pack(f'grad/param/{name}', p.grad, data, stride=7)
    head.eval()
    with torch.no_grad():
        dets = head.get_bboxes(*[[t.detach() for t in lv] for lv in outs], metas)
        raw = head.get_bboxes(*[[t.detach() for t in lv] for lv in outs], metas, nms=False)
    for i, (b, l) in enumerate(dets):
        data[f'det/{i}/bboxes'] = b.numpy()
        data[f'det/{i}/labels'] = l.numpy()
        gu.pack(f'raw/{i}/bboxes', raw[i][0], data)
    data['keys'] = np.array(sorted(head.state_dict().keys()))
    data['shapes'] = np.array([str(tuple(v.shape)) for _, v in sorted(head.state_dict().items())])
    _save('head_cpv', data)


def golden_decode_cpv():
    """Round 6: the REFERENCE's LSCPVHead.get_bboxes (lscpvnet_head.py:905-1090: per-level top-k, decode,
    corner verification against the stride-8 / stride-16 corner heat maps, multiclass NMS) on gu.cpv_decode_inputs -- head
    outputs derived from a seed alone, every hard selection away from a tie, so that the checking side holds labels, keep order
    and the max_per_img cut EXACTLY (tests/golden_cases.py cpv_decode_case)."""
    import mmcv
    from mmdet.models import build_head
    cfg, train_cfg, test_cfg = gu.cpv_head_cfg()
    cfg = mmcv.Config(copy.deepcopy(cfg))._cfg_dict
    cfg.update(train_cfg=mmcv.Config(train_cfg), test_cfg=mmcv.Config(test_cfg))
    head = build_head(cfg)
    gu.fill_params(head, seed=7)
    head.eval()
    data = {}
    with torch.no_grad():
        outs = head(gu.head_inputs(11))
        syn, gap = gu.cpv_decode_inputs(outs, 79)
        assert gap > 1e-3, gap
        _, _, _, _, _, metas = _gt_for('bbox')
        dets = head.get_bboxes(*syn, metas, cfg=mmcv.Config(gu.DECODE_CFG))
        raw = head.get_bboxes(*syn, metas, cfg=mmcv.Config(gu.DECODE_CFG), nms=False)
    data['min_gap'] = np.float64(gap)
    for i, (b, l) in enumerate(dets):
        print('cpv image', i, len(l), 'detections; top score', float(b[0, 4]) if len(l) else None)
        assert 20 < len(l) <= gu.DECODE_CFG['max_per_img']
        data[f'{i}/bboxes'] = b.numpy()
        data[f'{i}/labels'] = l.numpy()
        data[f'{i}/raw_bboxes'] = raw[i][0].numpy()       # every candidate's box after corner verification, before NMS
    _save('decode_cpv', data)


def golden_coco_eval():
    """(f-4) COCO metrics of the reference's vendored evaluator (cocoapi/pycocotools: coco.py loadRes, cocoeval.py, its
    maskApi.c bound by pycoco_mask.py) on gu.synthetic_eval_case: boxes, masks (polygons -> RLE), keypoints; also with
    `useCats = 0` and other detection budgets (the proposal-style use of mmdet/datasets/coco.py:430-445)."""
    import contextlib
    import io

    import pycocotools.mask as mask_util
    from pycocotools.coco import COCO
    from pycocotools.cocoeval import COCOeval
    gt_dict, boxes, polys, kpts = gu.synthetic_eval_case()
    data = {}
    sizes = {im['id']: (im['height'], im['width']) for im in gt_dict['images']}

    def run(kind, records, tweak=None):
        with contextlib.redirect_stdout(io.StringIO()):
            gt = COCO()
            gt.dataset = copy.deepcopy(gt_dict)
            gt.createIndex()
            dt = gt.loadRes(copy.deepcopy(records))
            ev = COCOeval(gt, dt, kind)
            if tweak:
                tweak(ev.params)
            ev.evaluate()
            ev.accumulate()
            ev.summarize()
        return ev
    segm = []
    for r in polys:
        h, w = sizes[r['image_id']]
        rle = mask_util.merge(mask_util.frPyObjects([r['polygon']], h, w))
        rle['counts'] = rle['counts'].decode()
        segm.append(dict(image_id=r['image_id'], category_id=r['category_id'], score=r['score'], segmentation=rle))
    data['segm/rle0'] = np.array(segm[0]['segmentation']['counts'])
    cases = dict(bbox=('bbox', boxes, None), segm=('segm', segm, None), keypoints=('keypoints', kpts, None),
                 bbox_nocat=('bbox', boxes, lambda p: (setattr(p, 'useCats', 0), setattr(p, 'maxDets', [3, 30, 300]))),
                 bbox_subset=('bbox', boxes, lambda p: (setattr(p, 'catIds', [1, 17]),
                                                        setattr(p, 'imgIds', [im['id'] for im in gt_dict['images'][:9]]))))
    for name, (kind, recs, tweak) in cases.items():
        ev = run(kind, recs, tweak)
        data[f'{name}/stats'] = np.asarray(ev.stats)
        data[f'{name}/precision'] = ev.eval['precision']
        data[f'{name}/recall'] = ev.eval['recall']
        data[f'{name}/scores'] = ev.eval['scores']
    _save('coco_eval', data)


CURVE_ITERS, CURVE_HW = 20, (288, 352)   # (a 20-iteration SGD curve with warm-up; 12 until round 4)


def curve_cfg():
    """Model / schedule of the training-curve fixture: configs/lsnet/lsnet_bbox_r50_fpn_1x_coco.py with the warm-up
    shortened to 6 iterations so that 12 iterations reach the full learning rate."""
    from lsnet_amd.model_zoo import lsnet_config
    cfg = lsnet_config('bbox', 'r50')
    cfg.lr_config = dict(policy='step', warmup='linear', warmup_iters=10, warmup_ratio=0.001, step=[14, 18])
    return cfg


def golden_train_curve(lr=0.01, fixture='train_curve', init0=False):
    """(8d) loss parity over an SGD run: the REFERENCE's detector, losses, optimizer hook (grad clip 35), SGD and
    step-LR warm-up hooks (mmcv runner, mmcv/runner/hooks/{optimizer,lr_updater}.py) for 12 iterations on one
    seeded synthetic batch per iteration, native ops backed by the CPU oracle.  Stores the loss curves, the learning
    rates and fingerprints of two weights after the run."""
    import logging

    import mmcv
    from mmcv.runner import EpochBasedRunner
    from mmdet.models import build_detector
    from lsnet_amd.data import synthetic_batch
    cfg = curve_cfg()
    model_cfg = mmcv.Config(copy.deepcopy(cfg.model.to_dict() if hasattr(cfg.model, 'to_dict') else dict(cfg.model)))._cfg_dict
    model = build_detector(model_cfg, train_cfg=mmcv.Config(dict(cfg.train_cfg)), test_cfg=mmcv.Config(dict(cfg.test_cfg)))
    if init0:
        # round 6: the UNTOUCHED seed-0 init_weights model -- the model `python bench.py` trains -- at
        # 2 x 3 x 384 x 512: the loss starts near 5.4 and moves smoothly, unlike the parameter-fill fixtures whose first iterations
        # are the collapse of a loss of 442
        from lsnet_amd.model_zoo import build_lsnet
        torch.manual_seed(0)
        own, _ = build_lsnet('bbox', 'r50')
        missing = model.load_state_dict(own.state_dict(), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    else:
        gu.fill_params(model, seed=11, head_norm_shift=0.0, pair_gap=1.0)   # (round 4's fill: golden_util.fill_params)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=0.0001)
    logger = logging.getLogger('curve')
    logger.setLevel(logging.ERROR)
    runner = EpochBasedRunner(model, optimizer=opt, work_dir=None, logger=logger)
    runner.register_training_hooks(dict(cfg.lr_config), dict(cfg.optimizer_config), None, dict(interval=10 ** 9, hooks=[]))
    if init0:
        batches = [synthetic_batch('bbox', 2, *gu.CURVE0_HW, num_classes=80, seed=700 + i, device='cpu', channels_last=False)
                   for i in range(CURVE_ITERS)]
    else:
        batches = [synthetic_batch('bbox', 1, *CURVE_HW, boxes_per_img=3, num_classes=80, seed=900 + i, device='cpu',
                                   channels_last=False) for i in range(CURVE_ITERS)]
    curves = {k: [] for k in ('loss', 'loss_cls', 'loss_bbox_init', 'loss_bbox_refine', 'lr')}

    class Tap(mmcv.runner.Hook):
        def after_train_iter(self, r):
            for k in curves:
                if k != 'lr':
                    curves[k].append(float(r.outputs['log_vars'][k]))
            curves['lr'].append(float(r.current_lr()[0]))
    runner.register_hook(Tap(), priority='VERY_LOW')
    runner.run([batches], [('train', 1)], 1)
    data = {f'curve/{k}': np.array(v, dtype=np.float64) for k, v in curves.items()}
    sd = model.state_dict()
    for name in ('bbox_head.pts_cls_out.weight', 'backbone.layer4.0.conv1.weight', 'neck.fpn_convs.0.conv.weight'):
        gu.pack(f'weight/{name}', sd[name], data, stride=211)
    _save(fixture, data)


def golden_bench_iter0():
    """SURVEY 8(d) "Loss parity" at the BENCHMARK shape: the reference's detector (mmdet LSDetector from
    configs/lsnet/lsnet_bbox_r50_fpn_1x_coco.py's model dict, native ops backed by the CPU oracle) on the very batch and
    the very weights `python bench.py` starts from -- torch.manual_seed(0), this package's model construction (the
    reference's init_weights semantics), synthetic_batch('bbox', 2, 800, 1344, seed=1234) -- one forward pass on the CPU.
    Stores the iteration-0 loss triplet and total; bench.py prints its own iteration-0 losses against them
    (`loss_ref_rel_err`) and tests/test_golden_gpu.py asserts 1e-3."""
    import mmcv
    from mmdet.models import build_detector
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet, lsnet_config
    torch.manual_seed(0)
    own, _ = build_lsnet('bbox', 'r50')
    cfg = lsnet_config('bbox', 'r50')
    model_cfg = mmcv.Config(copy.deepcopy(cfg.model.to_dict() if hasattr(cfg.model, 'to_dict') else dict(cfg.model)))._cfg_dict
    model = build_detector(model_cfg, train_cfg=mmcv.Config(dict(cfg.train_cfg)), test_cfg=mmcv.Config(dict(cfg.test_cfg)))
    missing = model.load_state_dict(own.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.train()
    data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device='cpu', channels_last=False)
    with torch.no_grad():
        losses = model(**data)
        _, log_vars = model._parse_losses(losses)
    out = {k: np.float64(float(v)) for k, v in log_vars.items()}
    out['per_level/loss_cls'] = np.array([float(v) for v in losses['loss_cls']], dtype=np.float64)
    out['per_level/loss_bbox_init'] = np.array([float(v) for v in losses['loss_bbox_init']], dtype=np.float64)
    out['per_level/loss_bbox_refine'] = np.array([float(v) for v in losses['loss_bbox_refine']], dtype=np.float64)
    print({k: v for k, v in out.items() if not k.startswith('per_level')})
    _save('bench_iter0', out)


def golden_assign():
    """(6) CentroidAssigner + ATSSAssigner gt indices on the 800x800 grid (13 343 points)."""
    from mmdet.core import build_assigner
    from mmdet.core.anchor.point_generator import PointGenerator
    data = {}
    strides = [8, 16, 32, 64, 128]
    sizes = [(-(-800 // s), -(-800 // s)) for s in strides]
    pg = PointGenerator()
    pts = torch.cat([pg.grid_points(sz, s, 'cpu') for sz, s in zip(sizes, strides)])
    init = build_assigner(dict(type='CentroidAssigner', scale=4, pos_num=1, iou_type='center'))
    cen = build_assigner(dict(type='CentroidAssigner', scale=4, pos_num=3, iou_type='centroid'))
    atss = build_assigner(dict(type='ATSSAssigner', topk=9))
    for i in range(2):
        b, l, e = gu.make_gt(300 + i, 7 + 5 * i, 800, 800)
        r = init.assign(pts, b, e, None, l)
        data[f'init/{i}/gt_inds'] = r.gt_inds.numpy().astype(np.int32)
        data[f'init/{i}/labels'] = r.labels.numpy().astype(np.int32)
        r = cen.assign(pts, b, e, None, l)
        data[f'centroid/{i}/gt_inds'] = r.gt_inds.numpy().astype(np.int32)
        g = gu.gen(400 + i)
        wh = torch.rand(pts.shape[0], 2, generator=g) * pts[:, 2:3] * 6 + 2
        ctr = pts[:, :2] + (torch.rand(pts.shape[0], 2, generator=g) - 0.5) * pts[:, 2:3]
        props = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
        r = atss.assign(props, [s[0] * s[1] for s in sizes], b, None, l)
        data[f'atss/{i}/gt_inds'] = r.gt_inds.numpy().astype(np.int32)
        gu.pack(f'atss/{i}/max_overlaps', r.max_overlaps, data, stride=5)
    _save('assign', data)


def golden_cross_iou():
    """(4) cross-IOU value + gradient for the bbox / polygon / keypoint variants, 4096 rows."""
    from mmdet.models.losses.cross_iou_loss import CrossIOULoss
    data = {}
    N = 4096
    for kind, width, nv in (('bbox', 20, 4), ('polygon', 148, 36), ('keypoint', 72, 17)):
        g = gu.gen(500 + width)
        pred = (torch.rand(N, width, generator=g) * 2 + 0.01).requires_grad_()
        target = torch.rand(N, width, generator=g) * 2
        pos = torch.rand(N, width // 2, generator=g) > 0.5
        pos = torch.stack([pos, ~pos], -1).reshape(N, width)
        target = target * pos            # the inactive half of each pair is 0 before alpha-filling
        rowpos = torch.rand(N, generator=g) < 0.02
        target = target * rowpos[:, None]
        weight = rowpos.float()[:, None].expand(N, width).contiguous()
        anchor = torch.rand(N, 2, generator=g) * 10
        lo = torch.rand(N, 2, generator=g) * 5
        bbox_gt = torch.cat([lo, lo + torch.rand(N, 2, generator=g) * 5 + 0.1], 1)
        vs = torch.randint(0, 3, (N, nv), generator=g).float()
        loss_fn = CrossIOULoss(loss_weight=2.0, loss_type=kind)
        kw = dict(anchor_pts=anchor, pos_inds=pos, bbox_gt=None if kind == 'keypoint' else bbox_gt)
        if kind == 'keypoint':
            kw['vs'] = vs.clone()
        loss = loss_fn(pred, target.clone(), weight, avg_factor=37.0, **kw)
        loss.backward()
        data[f'{kind}/loss'] = np.float64(loss.item())
        gu.pack(f'{kind}/grad', pred.grad, data, stride=3)
    _save('cross_iou', data)


def golden_backbone():
    """(9) ResNet-50 + FPN forward on a 1x3x128x128 input with name-keyed weights."""
    import mmcv
    from mmdet.models import build_backbone, build_neck
    cfg = mmcv.Config.fromfile('/root/reference/code/configs/lsnet/lsnet_bbox_r50_fpn_1x_coco.py')
    bb, neck = build_backbone(cfg.model.backbone), build_neck(cfg.model.neck)
    gu.fill_params(bb, seed=3); gu.fill_params(neck, seed=4)
    bb.train(); neck.train()
    x = torch.randn(1, 3, 128, 128, generator=gu.gen(21))
    data = {}
    with torch.no_grad():
        c = bb(x)
        p = neck(c)
    for i, t in enumerate(c):
        gu.pack(f'c/{i}', t, data)
    for i, t in enumerate(p):
        gu.pack(f'p/{i}', t, data)
    _save('backbone_r50_fpn', data)


def golden_res2net():
    """(f-3) Res2Net-50 (26w x 4s, DCNv2 in c3-c5: the structure of the headline res2_101 configs at a testable
    depth) forward and backward on a 1x3x96x128 input with name-keyed weights, plus the sorted state-dict keys."""
    from mmdet.models import build_backbone
    bb = build_backbone(dict(type='Res2Net', depth=50, scales=4, base_width=26, num_stages=4, out_indices=(0, 1, 2, 3),
                             frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True,
                             dcn=dict(type='DCNv2', deformable_groups=1, fallback_on_stride=False),
                             stage_with_dcn=(False, True, True, True)))
    gu.fill_params(bb, seed=8)
    bb.train()
    x = torch.randn(1, 3, 96, 128, generator=gu.gen(31)).requires_grad_()
    data = {'keys': np.array(sorted(bb.state_dict().keys())),
            'nparams': np.array(sum(p.numel() for p in bb.parameters()))}
    feats = bb(x)
    for i, t in enumerate(feats):
        gu.pack(f'c/{i}', t, data)
    # round 4: gradients of a fixed random projection of the four maps w.r.t. the image and parameters of every stage
    # (hierarchical 3x3 convs, the scale split / concat, average-pool shortcuts in backward)
    proj = sum((f * torch.randn(f.shape, generator=gu.gen(60 + i))).sum() / f.numel() ** 0.5 for i, f in enumerate(feats))
    names = gu.res2net_grad_names(bb)
    params = dict(bb.named_parameters())
    grads = torch.autograd.grad(proj, [x] + [params[n] for n in names])
    gu.pack('gx', grads[0], data)
    for n, g in zip(names, grads[1:]):
        gu.pack(f'g/{n}', g, data, stride=257)
    data['grad_names'] = np.array(names)
    _save('res2net50_dcn', data)


def golden_backbones_dcn():
    """BASELINE configs 3 / 4: ResNet-101 and ResNeXt-101-64x4d with DCNv2 in c3-c5 (g = 1 resp. g = 64, the first block
    of a stage with stride 2), full depth, 1x3x96x128 input, name-keyed weights: the four feature maps, and the
    gradients of a fixed random projection of them w.r.t. the input image and a few parameters of every stage (the
    backward of the grouped / strided deformable convs inside a real network)."""
    import copy
    import mmcv
    from mmdet.models import build_backbone
    sys.path.insert(0, '/root/repo')
    from lsnet_amd.model_zoo import backbone_cfg
    for name, fixture in (('r101-dcn', 'backbone_r101_dcn'), ('x101-dcn', 'backbone_x101_dcn')):
        cfg = backbone_cfg(name)
        cfg.pop('with_cp', None)
        bb = build_backbone(mmcv.Config(copy.deepcopy(cfg))._cfg_dict)
        bb.train()
        # (Round 6, VERDICT r5 item 5c -- "regenerate with kink-free inputs" -- tried and not possible: over 67 parameter-fill seeds
        # the closest ReLU pre-activation of the X-101 forward pass lay within 2e-8 .. 7e-8 of its tensor's range from zero.  A
        # 101-layer network at this input size always has gates inside rounding distance of the kink, whatever the seed; the
        # fixture keeps seed 13 and the test its measured bound: tests/golden_cases.py backbone_dcn_case.)
        gu.fill_params(bb, seed=13)
        x = torch.randn(1, 3, 96, 128, generator=gu.gen(41)).requires_grad_()
        data = {'keys': np.array(sorted(bb.state_dict().keys())),
                'nparams': np.array(sum(p.numel() for p in bb.parameters()))}
        feats = bb(x)
        proj = sum((f * torch.randn(f.shape, generator=gu.gen(50 + i))).sum() / f.numel() ** 0.5 for i, f in enumerate(feats))
        names = gu.backbone_grad_names(bb)
        params = dict(bb.named_parameters())
        grads = torch.autograd.grad(proj, [x] + [params[n] for n in names])
        for i, t in enumerate(feats):
            gu.pack(f'c/{i}', t, data)
        gu.pack('gx', grads[0], data)
        for n, g in zip(names, grads[1:]):
            gu.pack(f'g/{n}', g, data, stride=257)
        data['grad_names'] = np.array(names)
        _save(fixture, data)


def golden_vote():
    """(f-3) instances_vote of the reference detector (lsnet.py:229-299) on random multi-scale-like detections: clusters
    of jittered copies of a few boxes plus isolated ones."""
    from mmdet.models.detectors.lsnet import LSDetector
    g = gu.gen(91)
    data = {}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference ends with .cuda(); there is no GPU here
    try:
        for case, (nbase, ncopy, nv) in enumerate(((6, 7, 8), (15, 4, 72), (1, 1, 34), (3, 12, 34))):
            base = torch.rand(nbase, 2, generator=g) * 300
            wh = torch.rand(nbase, 2, generator=g) * 120 + 10
            b0 = torch.cat([base, base + wh], 1).repeat_interleave(ncopy, 0)
            boxes = b0 + torch.randn(b0.shape, generator=g) * 4
            vectors = torch.rand(boxes.shape[0], nv, generator=g) * 300
            scores = torch.rand(boxes.shape[0], generator=g)
            ob, ov, os_ = LSDetector.instances_vote(None, boxes.clone(), vectors.clone(), scores.clone())
            data[f'{case}/boxes'], data[f'{case}/vectors'], data[f'{case}/scores'] = boxes.numpy(), vectors.numpy(), scores.numpy()
            data[f'{case}/out_boxes'] = np.asarray(ob if isinstance(ob, np.ndarray) else ob.numpy(), dtype=np.float32).reshape(-1, 4)
            data[f'{case}/out_vectors'] = np.asarray(ov if isinstance(ov, np.ndarray) else ov.numpy(), dtype=np.float32).reshape(-1, nv)
            data[f'{case}/out_scores'] = np.asarray(os_ if isinstance(os_, np.ndarray) else os_.numpy(), dtype=np.float32).reshape(-1)
    finally:
        torch.Tensor.cuda = real_cuda
    _save('vote', data)


def golden_nms():
    """(7) multiclass_nms_lsvr keep set on random candidates (the reference's nms_cpu semantics)."""
    from mmdet.core import multiclass_nms_lsvr
    g = gu.gen(600)
    n, C = 1500, 8
    xy = torch.rand(n, 2, generator=g) * 300
    wh = torch.rand(n, 2, generator=g) * 80 + 2
    boxes = torch.cat([xy, xy + wh], 1)
    pts = torch.rand(n, 8, generator=g) * 300
    scores = torch.rand(n, C, generator=g) ** 4
    scores = torch.cat([scores, torch.zeros(n, 1)], 1)
    import mmcv
    dets, vec, labels = multiclass_nms_lsvr(boxes, pts, scores, 4, 0.05, mmcv.Config(dict(nms=dict(type='nms', iou_thr=0.6))).nms,
                                            100)
    _save('nms_lsvr', dict(dets=dets.numpy(), vectors=vec.numpy(), labels=labels.numpy()))


def golden_gt_formats():
    """(f-2) GT formatting of the real-data pipeline: polygon re-sampling / unification (loading.py:314-441) and the
    flip transforms (core/bbox/transforms.py:30-87) of the reference on random inputs."""
    from mmdet.core.bbox.transforms import extreme_flip, kps_flip, polygon_flip
    from mmdet.datasets.pipelines.loading import LoadAnnotations
    la = LoadAnnotations(with_mask=True, poly2mask=False, spline_num=10, num_contour_points=36)
    la.spline_poly_num = 36 * 10 if not hasattr(la, 'spline_poly_num') else la.spline_poly_num
    rng = np.random.RandomState(5)
    data = {}

    def blob(n, cx, cy, r):   # star-shaped, random orientation
        ang = np.sort(rng.rand(n)) * 2 * np.pi
        rad = r * (0.5 + rng.rand(n))
        pts = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
        return pts[::-1] if rng.rand() < 0.5 else pts
    cases = [[blob(4, 50, 60, 20)], [blob(7, 120, 80, 35)], [blob(57, 200, 150, 60)], [blob(400, 300, 200, 90)],
             [blob(12, 80, 90, 30), blob(5, 90, 95, 0.4)],            # second component is tiny: dropped
             [blob(3, 10, 10, 0.3)],                                  # everything tiny: box fallback
             [blob(9, 40, 40, 15), blob(30, 140, 60, 25)]]
    for i, comps in enumerate(cases):
        allp = np.concatenate(comps)
        bbox = np.array([allp[:, 0].min(), allp[:, 1].min(), allp[:, 0].max(), allp[:, 1].max()], dtype=np.float32)
        out = la.unify_polygons([c.reshape(-1).tolist() for c in comps], bbox)
        data[f'poly/{i}/n'] = np.array(len(comps))
        for j, c in enumerate(comps):
            data[f'poly/{i}/in{j}'] = c
        data[f'poly/{i}/bbox'] = bbox
        data[f'poly/{i}/out'] = np.stack(out)
    for n, m in ((5, 36), (400, 360), (361, 360), (3, 10), (13, 7)):
        pts = blob(n, 100, 100, 50)
        data[f'resample/{n}_{m}/in'] = pts
        data[f'resample/{n}_{m}/out'] = la.uniformsample(pts, m)
    g = gu.gen(77)
    shape = (480, 640, 3)
    ext = torch.rand(6, 16, generator=g) * 400
    pol = torch.rand(5, 72, generator=g) * 400
    kps = torch.rand(4, 34, generator=g) * 400
    data.update(ext=ext.numpy(), pol=pol.numpy(), kps=kps.numpy())
    for d in ('horizontal', 'vertical'):
        data[f'flip/{d}/ext'] = extreme_flip(ext, shape, d).numpy()
        data[f'flip/{d}/pol'] = polygon_flip(pol, shape, d).numpy()
        data[f'flip/{d}/kps'] = kps_flip(kps, shape, d).numpy()
    _save('gt_formats', data)


def tiny_coco(seed=3):
    """A small COCO-format annotation dict exercising every branch of the reference's annotation parser: polygon
    (one and several components, tiny components), crowd (run-length) and ignored instances, boxes outside the
    image / thinner than a pixel / of zero area, an unknown category, an image without annotations and one below
    the minimum size.  Carries both LSNet extras: `extreme_points` (10) and `keypoints` (51)."""
    rng = np.random.RandomState(seed)
    images = [dict(id=11, file_name='000011.jpg', width=97, height=64), dict(id=5, file_name='000005.jpg', width=60, height=90),
              dict(id=7, file_name='000007.jpg', width=40, height=36), dict(id=9, file_name='000009.jpg', width=20, height=200),
              dict(id=2, file_name='000002.jpg', width=128, height=96)]
    cats = [dict(id=1, name='person', supercategory='person'), dict(id=2, name='bicycle', supercategory='vehicle'),
            dict(id=18, name='dog', supercategory='animal'), dict(id=99, name='unicorn', supercategory='animal')]
    anns, aid = [], 100

    def star(n, cx, cy, r):
        ang = np.sort(rng.rand(n)) * 2 * np.pi
        rad = r * (0.5 + rng.rand(n))
        pts = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
        return (pts[::-1] if rng.rand() < 0.5 else pts).round(2)

    def add(img, cat, comps, crowd=0, ignore=None, bbox=None, area=None):
        nonlocal aid
        allp = np.concatenate(comps)
        x1, y1, x2, y2 = allp[:, 0].min(), allp[:, 1].min(), allp[:, 0].max(), allp[:, 1].max()
        box = [float(x1), float(y1), float(x2 - x1), float(y2 - y1)] if bbox is None else bbox
        k = np.concatenate([rng.rand(17, 2) * [x2 - x1, y2 - y1] + [x1, y1], rng.randint(0, 3, (17, 1))], 1).round(1)
        ex = np.array([[allp[allp[:, 1].argmin(), 0], y1], [x1, allp[allp[:, 0].argmin(), 1]],
                       [allp[allp[:, 1].argmax(), 0], y2], [x2, allp[allp[:, 0].argmax(), 1]],
                       [(x1 + x2) / 2, (y1 + y2) / 2]])
        seg = [c.reshape(-1).tolist() for c in comps]
        if crowd:
            seg = dict(size=[img['height'], img['width']], counts=[37, 120, int(img['height'] * img['width']) - 157])
        a = dict(id=aid, image_id=img['id'], category_id=cat, bbox=box, area=float(box[2] * box[3] * 0.6) if area is None else area,
                 iscrowd=crowd, segmentation=seg, extreme_points=ex.reshape(-1).round(2).tolist(),
                 keypoints=k.reshape(-1).tolist(), num_keypoints=int((k[:, 2] > 0).sum()))
        if ignore is not None:
            a['ignore'] = ignore
        anns.append(a)
        aid += 1
    im11, im5, im2 = images[0], images[1], images[4]
    add(im2, 1, [star(9, 40, 40, 18)])
    add(im11, 18, [star(7, 30, 30, 15)])
    add(im11, 1, [star(12, 60, 35, 20), star(5, 70, 40, 0.4)])
    add(im5, 2, [star(30, 30, 45, 22)])
    add(im11, 1, [star(6, 50, 30, 10)], crowd=1)
    add(im5, 1, [star(6, 20, 20, 8)], ignore=True)
    add(im5, 99, [star(6, 25, 60, 8)])
    add(im2, 1, [star(5, 64, 48, 12)], bbox=[200.0, 10.0, 20.0, 20.0])           # outside the image
    add(im2, 2, [star(5, 64, 48, 12)], bbox=[30.0, 30.0, 0.5, 20.0])             # thinner than a pixel
    add(im2, 18, [star(5, 64, 48, 12)], area=0.0)
    add(im2, 18, [star(40, 80, 50, 30), star(8, 20, 70, 9)])
    add(im2, 1, [star(3, 100, 20, 0.3)])                                         # all components tiny: box fallback
    add(im5, 1, [star(10, 35, 70, 14)])
    return dict(images=images, categories=cats, annotations=anns, info=dict(description='synthetic'), licenses=[])


def golden_data_pipeline():
    """(f-2) annotation parsing (datasets/coco.py:120-185, coco_pose.py:110-172) and the per-image pipeline
    (pipelines/loading.py LoadAnnotations, transforms.py Resize / RandomFlip / Pad, formating.py) of the reference on
    tests/golden/coco_tiny.json.  cv2 is absent from this image: its four calls behind mmcv (resize, cvtColor,
    subtract, multiply) are stood in by numpy here, so image PIXELS are not part of this fixture -- only shapes,
    meta data and every ground-truth field."""
    import json
    import types

    import cv2
    from lsnet_amd.data import geometry as G
    from mmdet.datasets.coco import CocoDataset
    from mmdet.datasets.coco_pose import CocoPoseDataset
    from mmdet.datasets.pipelines import Compose
    cv2.resize = lambda img, size, dst=None, interpolation=None: G.imresize(img, size)

    def _cvt(src, code, dst=None):
        dst[...] = src[..., ::-1].copy()
        return dst
    cv2.cvtColor, cv2.COLOR_BGR2RGB = _cvt, 4

    def _sub(a, b, dst=None):
        dst[...] = a - b.astype(np.float32)
        return dst

    def _mul(a, b, dst=None):
        dst[...] = a * b.astype(np.float32)
        return dst
    cv2.subtract, cv2.multiply = _sub, _mul

    coco = tiny_coco()
    with open(os.path.join(OUT, 'coco_tiny.json'), 'w') as f:
        json.dump(coco, f)
    data = {}
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    loaders = dict(bbox=(CocoDataset, dict(with_bbox=True, with_extreme=True), ['gt_bboxes', 'gt_labels', 'gt_extremes']),
                   segm=(CocoDataset, dict(with_bbox=True, with_mask=True, poly2mask=False, spline_num=10, num_contour_points=36),
                         ['gt_bboxes', 'gt_labels', 'gt_masks']),
                   pose=(CocoPoseDataset, dict(with_bbox=True, with_keypoint=True), ['gt_bboxes', 'gt_labels', 'gt_keypoints']))
    names = {c['id']: c['name'] for c in coco['categories']}
    for task, (cls, load_kw, keys) in loaders.items():
        classes = CocoDataset.CLASSES if cls is CocoDataset else ['person']
        cat_ids = [c['id'] for c in coco['categories'] if c['name'] in classes]
        stub = types.SimpleNamespace(cat_ids=cat_ids, cat2label={c: i for i, c in enumerate(cat_ids)})
        for img in coco['images']:
            info = dict(img, filename=img['file_name'])
            anns = [a for a in coco['annotations'] if a['image_id'] == img['id']]
            parsed = cls._parse_ann_info(stub, info, copy.deepcopy(anns))
            base = f'{task}/{img["id"]}'
            for k in ('bboxes', 'labels', 'bboxes_ignore', 'extremes', 'keypoints'):
                if k in parsed:
                    data[f'{base}/ann/{k}'] = np.asarray(parsed[k])
            data[f'{base}/ann/num_masks'] = np.array(len(parsed['masks']))
            if img['height'] < 32 or img['width'] < 32:
                continue
            for flip, direction in ((False, 'horizontal'), (True, 'horizontal'), (True, 'vertical')):
                pipe = Compose([dict(type='LoadAnnotations', **load_kw), dict(type='Resize', img_scale=(200, 120), keep_ratio=True),
                                dict(type='RandomFlip', flip_ratio=0.5), dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32),
                                dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img'] + keys)])
                rng = np.random.RandomState(img['id'])
                pixels = rng.randint(0, 256, (img['height'], img['width'], 3)).astype(np.uint8)
                results = dict(img_info=info, ann_info=copy.deepcopy(parsed), img=pixels, img_shape=pixels.shape, ori_shape=pixels.shape,
                               img_fields=['img'], filename=info['filename'], ori_filename=info['filename'], flip=flip,
                               flip_direction=direction, bbox_fields=[], extreme_fields=[], mask_fields=[], seg_fields=[],
                               keypoint_fields=[], img_prefix=None, seg_prefix=None, proposal_file=None)
                tag = f'{base}/{"flip_" + direction if flip else "plain"}'
                try:
                    out = pipe(results)
                except ValueError:            # the reference cannot mirror an EMPTY keypoint array (transforms.py:398)
                    assert task == 'pose' and len(parsed['keypoints']) == 0
                    data[f'{tag}/reference_raises'] = np.array(1)
                    continue
                meta = out['img_metas'].data
                data[f'{tag}/img_shape'] = np.array(meta['img_shape'])
                data[f'{tag}/pad_shape'] = np.array(meta['pad_shape'])
                data[f'{tag}/scale_factor'] = np.asarray(meta['scale_factor'])
                data[f'{tag}/img_tensor_shape'] = np.array(out['img'].data.shape)
                for k in keys:
                    v = out[k].data
                    if k == 'gt_masks':
                        data[f'{tag}/gt_masks/hw'] = np.array([v.height, v.width])
                        data[f'{tag}/gt_masks/ncomp'] = np.array([len(m) for m in v.masks])
                        flat = [c for m in v.masks for c in m]
                        data[f'{tag}/gt_masks/polys'] = np.stack(flat) if flat else np.zeros((0, 72))
                    else:
                        data[f'{tag}/{k}'] = v.numpy()
    # stride-8 semantic targets of the CPV variant (loading_reppointsv2.py:24-46)
    from mmdet.datasets.pipelines.loading_reppointsv2 import LoadRPDV2Annotations
    rng = np.random.RandomState(4)
    xy = rng.rand(12, 2) * [100, 70]
    boxes = np.concatenate([xy, xy + rng.rand(12, 2) * [60, 50] + 1], 1).astype(np.float32)
    boxes[:, 0::2] = boxes[:, 0::2].clip(0, 127)
    boxes[:, 1::2] = boxes[:, 1::2].clip(0, 95)
    labels = rng.randint(0, 5, 12).astype(np.int64)
    res = LoadRPDV2Annotations(num_classes=5)(dict(gt_bboxes=boxes, gt_labels=labels, pad_shape=(96, 128, 3)))
    data.update({'rpdv2/boxes': boxes, 'rpdv2/labels': labels, 'rpdv2/sem': res['gt_sem_map'], 'rpdv2/weights': res['gt_sem_weights']})
    # extreme points of tools/gen_coco_lsvr.py:16-75 on the polygon annotations and on random contours
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_coco_lsvr_ref', os.path.join(bootstrap.REF_ROOT, 'tools', 'gen_coco_lsvr.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    rng = np.random.RandomState(9)
    contours = [np.array([v for c in a['segmentation'] for v in c]).reshape(-1, 2) for a in coco['annotations']
                if isinstance(a['segmentation'], list)]
    for i in range(40):
        pts = (rng.rand(rng.randint(3, 40), 2) * rng.randint(5, 300)).round(rng.randint(0, 3))
        contours.append(pts.astype(np.int32) if i % 3 == 0 else pts)
    data['extreme/n'] = np.array(len(contours))
    for i, c in enumerate(contours):
        data[f'extreme/{i}/in'] = c
        data[f'extreme/{i}/out'] = tool._get_extreme_points(c)
    _save('data_pipeline', data)


ALL = dict(decode=golden_decode, decode_cpv=golden_decode_cpv, backbones_dcn=golden_backbones_dcn, bench_iter0=golden_bench_iter0, train_curve=golden_train_curve,
           # the same run at a tenth of the learning rate: the loss falls 475 -> 60 instead of 475 -> 1 and rounding
           # differences between two correct implementations stay at rounding level over all twelve iterations
           train_curve_lowlr=lambda: golden_train_curve(0.001, 'train_curve_lowlr'), train_curve_init0=lambda: golden_train_curve(0.01, 'train_curve_init0', init0=True), coco_eval=golden_coco_eval, head_cpv=golden_head_cpv, data_pipeline=golden_data_pipeline, gt_formats=golden_gt_formats, res2net=golden_res2net, vote=golden_vote, head_bbox=lambda: golden_head('bbox'), head_segm=lambda: golden_head('segm'),
           head_bbox_256=lambda: golden_head('bbox', 256),
           head_pose_bbox=lambda: golden_head('pose_bbox'), head_pose_kbox=lambda: golden_head('pose_kbox'),
           assign=golden_assign, cross_iou=golden_cross_iou, backbone=golden_backbone, nms=golden_nms)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    bootstrap.load_reference()
    torch.set_num_threads(8)
    for k in (args.only.split(',') if args.only else ALL):
        ALL[k]()
