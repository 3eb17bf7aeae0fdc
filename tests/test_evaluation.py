"""COCO evaluation: run-length masks of liblsnet_host.so and the AP / AR evaluator against
fixtures produced by the reference's vendored evaluator (tests/golden/coco_eval.npz, oracle/ref_harness/make_golden.py:
golden_coco_eval), against the reference's own maskApi.c where oracle/_ref holds it, and by their own properties."""
import copy
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from lsnet_amd.data import CocoDataset, CocoIndex, CocoPoseDataset
from lsnet_amd.evaluation import mask as mask_util
from lsnet_amd.evaluation.coco_eval import CocoEval, load_results
from tests import golden_util as gu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
ROOT = os.path.dirname(HERE)
REF_MASK = os.path.join(ROOT, 'oracle', '_ref', 'maskapi.so')


def test_host_library_exports_every_declared_symbol():
    import re
    with open(os.path.join(ROOT, 'include', 'lsnet_host.h')) as f:
        declared = set(re.findall(r'\b(lsn_\w+)\s*\(', f.read()))
    assert declared == set(mask_util.EXPORTS), declared ^ set(mask_util.EXPORTS)
    lib = ctypes.CDLL(os.path.join(ROOT, 'lsnet_amd', 'csrc', 'liblsnet_host.so'))
    for name in declared:
        assert hasattr(lib, name), name


def _square(x, y, s):
    return [x, y, x + s, y, x + s, y + s, x, y + s]


def test_run_length_mask_properties():
    h, w = 40, 60
    a = mask_util.frPyObjects([_square(10, 5, 20)], h, w)[0]
    m = mask_util.decode(a)
    assert m.shape == (h, w) and m.sum() == mask_util.area(a) == 400 and m[5:25, 10:30].all()
    assert mask_util.toBbox(a).tolist() == [10, 5, 20, 20]
    assert mask_util.encode(m) == a                                          # decode / encode round trip
    b = mask_util.frPyObjects([_square(20, 15, 20)], h, w)[0]
    uni, inter = mask_util.merge([a, b]), mask_util.merge([a, b], intersect=1)
    assert mask_util.area(inter) == 100 and mask_util.area(uni) == 700
    assert np.array_equal(mask_util.decode(uni), mask_util.decode(a) | mask_util.decode(b))
    iou = mask_util.iou([a, b], [b, uni], [0, 1])
    assert np.allclose(iou, [[100 / 700, 400 / 400], [1.0, 400 / 400]])      # crowd column: / detection area
    assert mask_util.iou([a], [], []) == [] and mask_util.iou([], [a], [0]) == []
    far = mask_util.frPyObjects([_square(45, 30, 5)], h, w)[0]
    assert mask_util.iou([a], [far], [0])[0, 0] == 0
    boxes = mask_util.iou(np.array([[0., 0, 10, 10]]), np.array([[5., 5, 10, 10], [20., 20, 5, 5]]), [0, 0])
    assert np.allclose(boxes, [[25 / 175, 0]])
    other = mask_util.frPyObjects([_square(10, 5, 20)], h + 1, w)[0]
    assert mask_util.iou([a], [other], [0])[0, 0] == -1                      # different grids
    multi = mask_util.frPyObjects([_square(1, 1, 3), _square(30, 30, 5)], h, w)
    assert [int(x) for x in mask_util.area(multi)] == [9, 25]
    s = a['counts']
    assert isinstance(s, bytes) and np.array_equal(mask_util.string_to_counts(s), [5 + 10 * h, 20] + [h - 20, 20] * 19 + [h * w - (29 * h + 25)])
    big = np.array([0, 5, 100000, 3, 99999, 4000000, 2, 7], dtype=np.uint32)
    assert np.array_equal(mask_util.string_to_counts(mask_util.counts_to_string(big)), big)
    # the dataset index decodes the same strings without the library
    from lsnet_amd.data.coco_index import rle_counts_from_string
    assert rle_counts_from_string(s.decode()) == mask_util.string_to_counts(s).tolist()


@pytest.mark.skipif(not os.path.exists(REF_MASK), reason='oracle/_ref/maskapi.so (built from the reference) is absent')
def test_run_length_masks_equal_reference_library():
    from oracle.ref_harness import pycoco_mask as ref
    rng = np.random.RandomState(0)

    def star(n, cx, cy, r):
        ang = np.sort(rng.rand(n)) * 2 * np.pi
        rad = r * (0.3 + rng.rand(n))
        return np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
    for i in range(300):                                                     # every grid, vertices inside and outside
        h, w = int(rng.randint(8, 200)), int(rng.randint(8, 200))
        p = star(int(rng.randint(3, 40)), rng.rand() * w, rng.rand() * h, rng.rand() * max(h, w) * 0.7)
        if i % 4 == 0:
            p = p.round()
        if i % 7 == 0:
            p[1] = p[0]                                                      # repeated vertex
        poly = p.reshape(-1).tolist()
        assert ref.frPyObjects([poly], h, w) == mask_util.frPyObjects([poly], h, w), i
    h, w = 120, 160
    polys = [star(int(rng.randint(3, 30)), rng.rand() * w, rng.rand() * h, 10 + rng.rand() * 60).reshape(-1).tolist()
             for _ in range(60)]
    R, M = ref.frPyObjects(polys, h, w), mask_util.frPyObjects(polys, h, w)
    assert R == M
    for i in range(0, 60, 3):
        for inter in (0, 1):
            assert ref.merge(R[i:i + 3], inter) == mask_util.merge(M[i:i + 3], inter)
    assert np.array_equal(ref.area(R), mask_util.area(M)) and np.array_equal(ref.toBbox(R), mask_util.toBbox(M))
    crowd = (rng.rand(25) < 0.3).astype(np.uint8)
    assert np.array_equal(ref.iou(R[:35], R[35:], crowd), mask_util.iou(M[:35], M[35:], crowd))
    b1, b2 = rng.rand(20, 4) * 50, rng.rand(30, 4) * 50
    c2 = (rng.rand(30) < 0.3).astype(np.uint8)
    assert np.array_equal(ref.iou(b1, b2, c2), mask_util.iou(b1, b2, c2))
    masks = (rng.rand(h, w, 3) < 0.5).astype(np.uint8)
    assert ref.encode(np.asfortranarray(masks)) == mask_util.encode(masks)
    assert np.array_equal(ref.decode(R[:4]), mask_util.decode(M[:4]))
    bx = (rng.rand(10, 4) * 60).round(1)
    assert ref.frPyObjects(bx, h, w) == mask_util.frPyObjects(bx, h, w)
    unc = [dict(size=[h, w], counts=[5, 10, h * w - 15])]
    assert ref.frPyObjects(unc, h, w) == mask_util.frPyObjects(unc, h, w)


def _evaluate(kind, gt_dict, records, tweak=None):
    gt = CocoIndex(dataset=copy.deepcopy(gt_dict))
    ev = CocoEval(gt, load_results(gt, copy.deepcopy(records)), kind)
    if tweak:
        tweak(ev.params)
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    return ev


def _segm_records(gt_dict, polys):
    sizes = {im['id']: (im['height'], im['width']) for im in gt_dict['images']}
    out = []
    for r in polys:
        h, w = sizes[r['image_id']]
        rle = mask_util.merge(mask_util.frPyObjects([r['polygon']], h, w))
        rle['counts'] = rle['counts'].decode()
        out.append(dict(image_id=r['image_id'], category_id=r['category_id'], score=r['score'], segmentation=rle))
    return out


@pytest.mark.parametrize('case', ['bbox', 'segm', 'keypoints', 'bbox_nocat', 'bbox_subset'])
def test_metrics_equal_reference_evaluator(case):
    gold = np.load(os.path.join(GOLD, 'coco_eval.npz'))
    gt_dict, boxes, polys, kpts = gu.synthetic_eval_case()
    tweak = None
    if case == 'bbox_nocat':
        def tweak(p):
            p.use_cats, p.max_dets = 0, [3, 30, 300]
    elif case == 'bbox_subset':
        def tweak(p):
            p.cat_ids, p.img_ids = [1, 17], [im['id'] for im in gt_dict['images'][:9]]
    if case == 'segm':
        records = _segm_records(gt_dict, polys)
        assert records[0]['segmentation']['counts'] == str(gold['segm/rle0'])      # polygon -> RLE, as the reference's
    else:
        records = kpts if case == 'keypoints' else boxes
    ev = _evaluate(case.split('_')[0], gt_dict, records, tweak)
    assert ev.eval['precision'].shape == gold[f'{case}/precision'].shape
    np.testing.assert_allclose(ev.eval['precision'], gold[f'{case}/precision'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ev.eval['recall'], gold[f'{case}/recall'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ev.eval['scores'], gold[f'{case}/scores'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ev.stats, gold[f'{case}/stats'], rtol=0, atol=1e-12)
    assert (ev.stats[:3] > 0).any()


def test_perfect_and_empty_detections():
    gt_dict, *_ = gu.synthetic_eval_case(seed=1, num_images=6)
    perfect = [dict(image_id=a['image_id'], category_id=a['category_id'], bbox=list(a['bbox']), score=0.9)
               for a in gt_dict['annotations'] if not a['iscrowd']]
    ev = _evaluate('bbox', gt_dict, perfect)
    assert ev.stats[0] == pytest.approx(1.0) and ev.stats[8] == pytest.approx(1.0)
    with pytest.raises(IndexError):
        load_results(CocoIndex(dataset=gt_dict), [])
    with pytest.raises(AssertionError):
        load_results(CocoIndex(dataset=gt_dict), [dict(image_id=-5, category_id=1, bbox=[0, 0, 1, 1], score=1.0)])


def test_dataset_evaluate_from_detector_outputs(tmp_path):
    """`CocoDataset.evaluate` on results in the detectors' output form: boxes (bbox task / CPV), (boxes, RLEs) for
    the segm task through `encode_poly_results`, [boxes, keypoints] for the pose tasks."""
    from lsnet_amd.apis import encode_poly_results
    gt_dict, boxes, polys, kpts = gu.synthetic_eval_case()
    for a in gt_dict['annotations']:
        a['extreme_points'] = [0.0] * 10
    gt_dict['categories'] = [dict(c, name={'person': 'person', 'car': 'car', 'cat': 'cat'}[c['name']]) for c in gt_dict['categories']]
    path = tmp_path / 'gt.json'
    path.write_text(json.dumps(gt_dict))
    ds = CocoDataset(str(path), pipeline=[], test_mode=True)
    assert ds.cat_ids == [1, 3, 17]
    label = {c: i for i, c in enumerate(ds.cat_ids)}
    gold = np.load(os.path.join(GOLD, 'coco_eval.npz'))

    def per_image(records, width, key=None):
        res = []
        for img_id in ds.img_ids:
            cls = [[] for _ in ds.cat_ids]
            for r in records:
                if r['image_id'] == img_id:
                    x, y, w, h = r['bbox'] if 'bbox' in r else (0, 0, 1, 1)
                    cls[label[r['category_id']]].append((np.array([x, y, x + w, y + h, r['score']], dtype=np.float64),
                                                         None if key is None else np.asarray(r[key], dtype=np.float64)))
            res.append(cls)
        return res
    grouped = per_image(boxes, 5)
    results = [[np.stack([b for b, _ in c]) if c else np.zeros((0, 5)) for c in cls] for cls in grouped]
    out = ds.evaluate(results, metric='bbox', classwise=True)
    # float round trip through xyxy -> xywh perturbs boxes in the last bit: headline numbers to 3 decimals
    for i, item in enumerate(('mAP', 'mAP_50', 'mAP_75', 'mAP_s', 'mAP_m', 'mAP_l')):
        assert out[f'bbox_{item}'] == pytest.approx(float(f'{gold["bbox/stats"][i]:.3f}'), abs=2e-3)
    assert set(out['bbox_classwise_AP']) == {'person', 'car', 'cat'} and len(out['bbox_mAP_copypaste'].split()) == 6

    # segm: polygons of each detection + its bounding box
    sizes = {im['id']: (im['height'], im['width']) for im in gt_dict['images']}
    seg_results = []
    for img_id in ds.img_ids:
        cls_b, cls_p = [[] for _ in ds.cat_ids], [[] for _ in ds.cat_ids]
        for r in polys:
            if r['image_id'] == img_id:
                p = np.asarray(r['polygon']).reshape(-1, 2)
                cls_b[label[r['category_id']]].append([p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max(), r['score']])
                cls_p[label[r['category_id']]].append(np.asarray(r['polygon']))
        h, w = sizes[img_id]
        seg_results.append(([np.array(b, dtype=np.float64).reshape(-1, 5) for b in cls_b], encode_poly_results(cls_p, h, w)))
    out = ds.evaluate(seg_results, metric=['bbox', 'segm'])
    # records with a 'bbox' take their area from the box (coco.py loadRes), the fixture's took it from the mask: only the
    # size-independent numbers are comparable
    for i, item in enumerate(('mAP', 'mAP_50', 'mAP_75')):
        assert out[f'segm_{item}'] == float(f'{gold["segm/stats"][i]:.3f}')
    assert all(-1 <= out[f'segm_{k}'] <= 1 for k in ('mAP_s', 'mAP_m', 'mAP_l'))
    assert 'bbox_mAP' in out

    # pose: [boxes, keypoints (x, y pairs)] per image; every keypoint reported visible
    pose = CocoPoseDataset(str(path), pipeline=[], test_mode=True)
    pose_results = []
    for img_id in pose.img_ids:
        b, k = [], []
        for r in kpts:
            if r['image_id'] == img_id:
                x, y, w, h = r['bbox']
                b.append([x, y, x + w, y + h, r['score']])
                k.append(np.asarray(r['keypoints']).reshape(17, 3)[:, :2].reshape(-1))
        pose_results.append([[np.array(b, dtype=np.float64).reshape(-1, 5)], [np.array(k, dtype=np.float32).reshape(-1, 34)]])
    out = pose.evaluate(pose_results, metric='keypoints')
    def person_only(p):
        p.cat_ids = [1]
    direct = _evaluate('keypoints', gt_dict, kpts, person_only)       # the fixture scored all three categories
    for i, item in enumerate(('mAP', 'mAP_50', 'mAP_75', 'mAP_s', 'mAP_m', 'mAP_l')):      # the reference's key names
        assert out[f'keypoints_{item}'] == pytest.approx(float(f'{direct.stats[i]:.3f}'), abs=2e-3)
    assert out['keypoints_mAP'] > 0
    with pytest.raises(KeyError):
        ds.evaluate(results, metric='proposal')
    assert torch is not None


def _collect_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from lsnet_amd.apis import collect_results
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        part = [dict(idx=i, arr=np.full(3, i)) for i in range(rank, 8, world)]       # 7 samples padded to 8
        got = collect_results(part, 7)
        if rank == 0:
            assert [g['idx'] for g in got] == list(range(7)) and all((g['arr'] == g['idx']).all() for g in got)
            open(os.path.join(out_dir, 'ok'), 'w').close()
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_collect_results_two_ranks(tmp_path):
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_collect_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok')


@pytest.mark.parametrize('task', ['segm'])          # the bbox flavour runs inside test_train_detector_with_validation
def test_test_loop_and_evaluate(tmp_path, task, cpu_oracle_backend):
    """test loader -> single_gpu_test (LSNet R-50, random weights) -> CocoDataset.evaluate: plumbing end to end."""
    from lsnet_amd.apis import single_gpu_test
    from lsnet_amd.data import build_dataloader, build_dataset
    from lsnet_amd.model_zoo import build_lsnet
    from tests.test_data_pipeline import NORM, _write_images
    from lsnet_amd.data.datasets import COCO_CLASSES
    ann = _write_images(str(tmp_path))
    with open(ann) as f:
        coco = json.load(f)
    have = {c['name'] for c in coco['categories']}
    coco['categories'] += [dict(id=200 + i, name=n, supercategory='x') for i, n in enumerate(COCO_CLASSES) if n not in have]
    with open(ann, 'w') as f:
        json.dump(coco, f)                                      # all 80 classes exist, as in the real annotation files
    ds = build_dataset(dict(type='CocoDataset', ann_file=ann, img_prefix=str(tmp_path), test_mode=True, pipeline=[
        dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=(480, 384), flip=False, transforms=[
            dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), dict(type='Normalize', **NORM),
            dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]))
    loader = build_dataloader(ds, samples_per_gpu=1, workers_per_gpu=0, dist=False, shuffle=False)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, _ = build_lsnet(task, 'r50')
    model.test_cfg.nms_pre, model.test_cfg.max_per_img, model.test_cfg.score_thr = 20, 10, 0.0
    results = single_gpu_test(model, loader)
    assert len(results) == len(ds) == 5
    if task == 'bbox':
        assert all(len(r) == 80 and all(c.shape[1] == 5 for c in r) for r in results)
        out = ds.evaluate(results, metric='bbox')
        assert -1 <= out['bbox_mAP'] <= 1
    else:
        boxes, rles = results[0]
        assert len(boxes) == len(rles) == 80
        k = next(i for i, c in enumerate(boxes) if len(c))
        info = ds.data_infos[0]
        assert rles[k][0]['size'] == [info['height'], info['width']] and len(rles[k]) == len(boxes[k])
        out = ds.evaluate(results, metric=['bbox', 'segm'])
        assert -1 <= out['segm_mAP'] <= 1 and 'bbox_mAP' in out


def test_train_detector_with_validation(tmp_path, cpu_oracle_backend):
    """`train_detector(validate=True)`: the EvalHook tests the model on `cfg.data.val` after the epoch and scores it
    with `cfg.evaluation` (mmdet/apis/train.py:112-122, core/evaluation/eval_hooks.py)."""
    from lsnet_amd.apis import train_detector
    from lsnet_amd.data import build_dataloader, build_dataset
    from lsnet_amd.data.datasets import COCO_CLASSES
    from lsnet_amd.model_zoo import build_lsnet
    from tests.test_data_pipeline import NORM, TASKS, _pipeline, _write_images
    ann = _write_images(str(tmp_path))
    with open(ann) as f:
        coco = json.load(f)
    have = {c['name'] for c in coco['categories']}
    coco['categories'] += [dict(id=200 + i, name=n, supercategory='x') for i, n in enumerate(COCO_CLASSES) if n not in have]
    with open(ann, 'w') as f:
        json.dump(coco, f)
    cls, load_kw, keys = TASKS['bbox']
    train = build_dataset(dict(type='CocoDataset', ann_file=ann, img_prefix=str(tmp_path),
                               pipeline=[dict(type='LoadImageFromFile')] + _pipeline(load_kw, keys, scale=(480, 384))))
    torch.manual_seed(1)
    torch.set_num_threads(8)
    model, cfg = build_lsnet('bbox', 'r50')
    model.test_cfg.nms_pre, model.test_cfg.max_per_img, model.test_cfg.score_thr = 20, 10, 0.0
    cfg.total_epochs, cfg.workflow, cfg.checkpoint_config = 1, [('train', 1)], None
    cfg.log_config = dict(interval=10 ** 9, hooks=[])
    cfg.data.workers_per_gpu = 0
    cfg.data.val = dict(type='CocoDataset', ann_file=ann, img_prefix=str(tmp_path), pipeline=[
        dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=(480, 384), flip=False, transforms=[
            dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), dict(type='Normalize', **NORM),
            dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])])
    cfg.evaluation = dict(interval=1, metric=['bbox'])
    lines = []
    before = model.bbox_head.pts_cls_out.weight.detach().clone()
    loader = build_dataloader(train, 1, 0, dist=False, shuffle=True, seed=0)
    runner = train_detector(model, [loader], cfg, distributed=False, validate=True, logger=lines.append, channels_last=False)
    assert runner.iter == len(loader) == 3 and not torch.equal(before, model.bbox_head.pts_cls_out.weight)
    loss = float(runner.outputs['log_vars']['loss'])
    assert np.isfinite(loss) and loss > 0
    assert runner.epoch == 1 and -1 <= runner.eval_results['bbox_mAP'] <= 1
    assert any('Epoch(val) [1]' in str(s) and 'bbox_mAP' in str(s) for s in lines)
    assert model.training
