"""Samplers of the real-data pipeline.  Where /root/reference is present the indices are compared with the reference's
own DistributedGroupSampler; everywhere, with the invariants that define it."""
import os
import types

import numpy as np
import pytest

from lsnet_amd.data.samplers import DistributedGroupSampler


def _dataset(n, seed):
    rng = np.random.RandomState(seed)
    return types.SimpleNamespace(flag=(rng.rand(n) < 0.7).astype(np.uint8), __len__=lambda: n)


@pytest.mark.parametrize('n,spg,world', [(103, 2, 8), (64, 4, 2), (37, 3, 1), (10, 2, 4)])
def test_group_sampler_invariants(n, spg, world):
    ds = _dataset(n, n)
    per_rank = []
    for r in range(world):
        s = DistributedGroupSampler(ds, spg, world, r)
        s.set_epoch(3)
        idx = list(s)
        assert len(idx) == len(s) and len(idx) % spg == 0
        for b in range(0, len(idx), spg):                       # one group per mini-batch
            assert len({int(ds.flag[i]) for i in idx[b:b + spg]}) == 1
        per_rank.append(idx)
    assert len({len(p) for p in per_rank}) == 1                 # same number of batches on every rank
    seen = set(i for p in per_rank for i in p)
    assert seen == set(range(n))                                # every image is used (padding repeats some)
    s = DistributedGroupSampler(ds, spg, world, 0)
    s.set_epoch(4)
    assert list(s) != per_rank[0]                               # another epoch, another order


@pytest.mark.skipif(not os.path.isdir('/root/reference/code'), reason='the reference tree is not on this machine')
def test_group_sampler_equals_reference():
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    from oracle.ref_harness import bootstrap
    bootstrap.load_reference()
    from mmdet.datasets.samplers import DistributedGroupSampler as Ref
    for n, spg, world in ((103, 2, 8), (64, 4, 2), (37, 3, 1)):
        ds = _dataset(n, n)
        for r in range(world):
            a, b = DistributedGroupSampler(ds, spg, world, r), Ref(ds, spg, world, r)
            for ep in (0, 5):
                a.set_epoch(ep)
                b.set_epoch(ep)
                assert list(a) == [int(i) for i in b], (n, spg, world, r, ep)


# ---------------------------------------------------------------------------------------------------------------------
# annotation parsing + per-image pipeline against fixtures produced by the reference's own classes
# (oracle/ref_harness/make_golden.py: golden_data_pipeline) on tests/golden/coco_tiny.json
import copy  # noqa: E402
import json  # noqa: E402

import torch  # noqa: E402

from lsnet_amd.data import (CocoDataset, CocoPoseDataset, Compose, GroupSampler, PolygonMasks, build_dataloader,  # noqa: E402
                            build_dataset, rle_decode)
from lsnet_amd.data import geometry as G  # noqa: E402
from lsnet_amd.data.coco_index import rle_counts_from_string  # noqa: E402
from lsnet_amd.parallel.data_container import DataContainer, collate, scatter  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TINY = os.path.join(GOLD, 'coco_tiny.json')
NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
TASKS = dict(bbox=(CocoDataset, dict(with_bbox=True, with_extreme=True), ['gt_bboxes', 'gt_labels', 'gt_extremes']),
             segm=(CocoDataset, dict(with_bbox=True, with_mask=True, poly2mask=False, spline_num=10, num_contour_points=36),
                   ['gt_bboxes', 'gt_labels', 'gt_masks']),
             pose=(CocoPoseDataset, dict(with_bbox=True, with_keypoint=True), ['gt_bboxes', 'gt_labels', 'gt_keypoints']))


def _pipeline(load_kw, keys, scale=(200, 120)):
    return [dict(type='LoadAnnotations', **load_kw), dict(type='Resize', img_scale=scale, keep_ratio=True),
            dict(type='RandomFlip', flip_ratio=0.5), dict(type='Normalize', **NORM), dict(type='Pad', size_divisor=32),
            dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img'] + keys)]


def test_dataset_index_filter_and_group_flags():
    ds = CocoDataset(TINY, pipeline=[])
    assert [d['id'] for d in ds.data_infos] == [11, 5, 2]          # 7: no annotations, 9: no annotations and too small
    assert ds.flag.tolist() == [1, 0, 1]
    assert ds.cat_ids == [1, 2, 18] and ds.cat2label == {1: 0, 2: 1, 18: 2}
    full = CocoDataset(TINY, pipeline=[], test_mode=True)
    assert [d['id'] for d in full.data_infos] == [11, 5, 7, 9, 2] and not hasattr(full, 'flag')
    keep = CocoDataset(TINY, pipeline=[], filter_empty_gt=False)
    assert [d['id'] for d in keep.data_infos] == [11, 5, 7, 2]     # 9 is narrower than 32 px
    pose = CocoPoseDataset(TINY, pipeline=[])
    assert pose.cat_ids == [1] and pose.get_cat_ids(0) == [18, 1, 1]


@pytest.mark.parametrize('task', list(TASKS))
def test_annotations_and_pipeline_equal_reference_fixture(task):
    gold = np.load(os.path.join(GOLD, 'data_pipeline.npz'))
    cls, load_kw, keys = TASKS[task]
    ds = cls(TINY, pipeline=_pipeline(load_kw, keys), test_mode=True)    # test_mode: keep every image, as the fixture does
    checked = 0
    for idx, info in enumerate(ds.data_infos):
        base = f'{task}/{info["id"]}'
        ann = ds.get_ann_info(idx)
        for k in ('bboxes', 'labels', 'bboxes_ignore', 'extremes', 'keypoints'):
            if k in ann:
                ref = gold[f'{base}/ann/{k}']
                assert ann[k].dtype == ref.dtype and ann[k].shape == ref.shape and np.array_equal(ann[k], ref), (base, k)
        assert len(ann['masks']) == int(gold[f'{base}/ann/num_masks'])
        if min(info['width'], info['height']) < 32:
            continue
        for flip, direction in ((False, 'horizontal'), (True, 'horizontal'), (True, 'vertical')):
            tag = f'{base}/{"flip_" + direction if flip else "plain"}'
            pixels = np.random.RandomState(info['id']).randint(0, 256, (info['height'], info['width'], 3)).astype(np.uint8)
            results = dict(img_info=info, ann_info=copy.deepcopy(ann), img=pixels, img_shape=pixels.shape,
                           ori_shape=pixels.shape, img_fields=['img'], filename=info['filename'],
                           ori_filename=info['filename'], flip=flip, flip_direction=direction)
            ds.pre_pipeline(results)
            out = ds.pipeline(results)
            if f'{tag}/reference_raises' in gold:            # empty keypoints + mirror: the reference throws, we return empty
                assert out['gt_keypoints'].data.shape == (0, 51)
                continue
            meta = out['img_metas'].data
            assert tuple(meta['img_shape']) == tuple(gold[f'{tag}/img_shape'])
            assert tuple(meta['pad_shape']) == tuple(gold[f'{tag}/pad_shape'])
            assert np.array_equal(meta['scale_factor'], gold[f'{tag}/scale_factor'])
            assert meta['flip'] == flip and meta['flip_direction'] == direction
            img = out['img']
            assert img.stack and tuple(img.data.shape) == tuple(gold[f'{tag}/img_tensor_shape']) and img.data.dtype == torch.float32
            for k in keys:
                if k == 'gt_masks':
                    m = out[k]
                    assert m.cpu_only and isinstance(m.data, PolygonMasks)
                    assert [m.data.height, m.data.width] == gold[f'{tag}/gt_masks/hw'].tolist()
                    assert [len(o) for o in m.data.masks] == gold[f'{tag}/gt_masks/ncomp'].tolist()
                    flat = [c for o in m.data.masks for c in o]
                    got = np.stack(flat) if flat else np.zeros((0, 72))
                    np.testing.assert_allclose(got, gold[f'{tag}/gt_masks/polys'], rtol=1e-6, atol=1e-4)
                else:
                    ref = gold[f'{tag}/{k}']
                    got = out[k].data.numpy()
                    assert got.dtype == ref.dtype and got.shape == ref.shape, (tag, k)
                    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-5, err_msg=f'{tag}/{k}')
            checked += 1
    assert checked >= 6


# ---------------------------------------------------------------------------------------------------------------------
# image helpers (cv2 stand-ins): pinned by properties
def test_rescale_size_and_padding():
    assert G.rescale_size((640, 480), (1333, 800)) == (1067, 800)
    assert G.rescale_size((480, 640), (1333, 800)) == (800, 1067)
    assert G.rescale_size((2000, 500), (1333, 800)) == (1333, 333)
    assert G.rescale_size((100, 50), 1.5) == (150, 75)
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    p = G.impad_to_multiple(img, 4, pad_val=9)
    assert p.shape == (8, 8, 3) and np.array_equal(p[:5, :7], img) and (p[5:] == 9).all() and (p[:, 7:] == 9).all()
    assert G.impad_to_multiple(p, 4) is not p and G.impad_to_multiple(p, 4).shape == (8, 8, 3)


def test_bilinear_resize_properties():
    rng = np.random.RandomState(0)
    const = np.full((13, 17, 3), 201, dtype=np.uint8)
    assert (G.imresize(const, (40, 31)) == 201).all()                       # weights sum to one, also in fixed point
    img = rng.randint(0, 256, (24, 36, 3)).astype(np.uint8)
    assert np.array_equal(G.imresize(img, (36, 24)), img)
    up = G.imresize(img, (72, 48))
    ref = G.imresize(img.astype(np.float32), (72, 48))
    assert up.dtype == np.uint8 and np.abs(up.astype(np.float32) - ref).max() <= 1.0       # fixed point vs float path
    # exact 2x up-sampling: taps at -0.25/+0.25 around source centres, borders clamp
    row = np.array([[0, 100, 200]], dtype=np.float32).repeat(2, 0)[..., None]
    out = G.imresize(row, (6, 2))[0, :, 0]
    np.testing.assert_allclose(out, [0, 25, 75, 125, 175, 200], atol=1e-4)
    # torch's bilinear (align_corners=False, no antialias) is the same sampling rule
    t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    tt = torch.nn.functional.interpolate(t, size=(17, 50), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(G.imresize(img.astype(np.float32), (50, 17)), tt, atol=2e-3)
    near = G.imresize(img, (18, 12), interpolation='nearest')
    assert np.array_equal(near, img[::2, ::2])


def test_normalize_flip_and_decode(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(1)
    bgr = rng.randint(0, 256, (9, 11, 3)).astype(np.uint8)
    out = G.imnormalize(bgr, np.array(NORM['mean'], np.float32), np.array(NORM['std'], np.float32), True)
    ref = (bgr[..., ::-1].astype(np.float64) - np.array(NORM['mean'])) / np.array(NORM['std'])
    assert out.dtype == np.float32 and out.flags['C_CONTIGUOUS']
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6)
    assert np.array_equal(G.imflip(bgr), bgr[:, ::-1]) and np.array_equal(G.imflip(bgr, 'vertical'), bgr[::-1])
    Image.fromarray(bgr[..., ::-1]).save(tmp_path / 'a.png')
    assert np.array_equal(G.imread(str(tmp_path / 'a.png')), bgr)          # decoded to BGR


def test_run_length_masks():
    rng = np.random.RandomState(2)
    mask = (rng.rand(13, 9) < 0.4).astype(np.uint8)
    flat = mask.T.reshape(-1)                                               # column-major runs, starting with zeros
    counts, cur, run = [], 0, 0
    for v in flat:
        if v == cur:
            run += 1
        else:
            counts.append(run)
            cur, run = v, 1
    counts.append(run)
    assert np.array_equal(rle_decode(dict(size=[13, 9], counts=counts)), mask)

    def encode(cnts):                                                       # inverse of the string form, for the round trip
        s = bytearray()
        for i, c in enumerate(cnts):
            x = c - cnts[i - 2] if i > 2 else c
            more = True
            while more:
                ch = x & 0x1f
                x >>= 5
                more = not ((x == 0 and not (ch & 0x10)) or (x == -1 and (ch & 0x10)))
                s.append((ch | (0x20 if more else 0)) + 48)
        return bytes(s).decode()
    big = [0, 5, 1000, 3, 999, 40000, 2, 7]
    assert rle_counts_from_string(encode(big)) == big
    assert np.array_equal(rle_decode(dict(size=[13, 9], counts=encode(counts))), mask)


# ---------------------------------------------------------------------------------------------------------------------
# collate / scatter and the loader end to end
def test_collate_and_scatter():
    def sample(h, w, n):
        return dict(img=DataContainer(torch.full((3, h, w), float(n)), stack=True),
                    gt_bboxes=DataContainer(torch.ones(n, 4) * n), img_metas=DataContainer(dict(n=n), cpu_only=True),
                    gt_masks=DataContainer(PolygonMasks([[np.zeros(72)]] * n, h, w), cpu_only=True))
    batch = collate([sample(32, 64, 1), sample(64, 32, 2), sample(96, 96, 3), sample(32, 32, 4)], samples_per_gpu=2)
    assert [tuple(t.shape) for t in batch['img'].data] == [(2, 3, 64, 64), (2, 3, 96, 96)]
    first = batch['img'].data[0]
    assert (first[0, :, :32, :64] == 1).all() and (first[0, :, 32:] == 0).all() and (first[1, :, :, 32:] == 0).all()
    assert [len(g) for g in batch['gt_bboxes'].data] == [2, 2] and batch['gt_bboxes'].data[1][1].shape == (4, 4)
    assert batch['img_metas'].cpu_only and batch['img_metas'].data[1] == [dict(n=3), dict(n=4)]
    with pytest.raises(AssertionError):
        collate([sample(32, 32, 1)] * 3, samples_per_gpu=2)
    one = scatter(collate([sample(32, 64, 1), sample(64, 32, 2)], samples_per_gpu=2), device='cpu')
    assert isinstance(one['img'], torch.Tensor) and one['img'].shape == (2, 3, 64, 64)
    assert isinstance(one['gt_bboxes'], list) and one['gt_bboxes'][1].shape == (2, 4)
    assert one['img_metas'] == [dict(n=1), dict(n=2)] and len(one['gt_masks'][1]) == 2


def _write_images(root):
    from PIL import Image
    with open(TINY) as f:
        coco = json.load(f)
    for im in coco['images']:
        px = np.random.RandomState(im['id']).randint(0, 256, (im['height'], im['width'], 3)).astype(np.uint8)
        Image.fromarray(px).save(os.path.join(root, im['file_name'].replace('jpg', 'png')))
    for im in coco['images']:
        im['file_name'] = im['file_name'].replace('jpg', 'png')
    path = os.path.join(root, 'ann.json')
    with open(path, 'w') as f:
        json.dump(coco, f)
    return path


@pytest.mark.parametrize('workers', [0, 2])
def test_loader_end_to_end(tmp_path, workers):
    ann = _write_images(str(tmp_path))
    cls, load_kw, keys = TASKS['segm']
    cfg = dict(type='RepeatDataset', times=3,
               dataset=dict(type='CocoDataset', ann_file='ann.json', data_root=str(tmp_path), img_prefix='',
                            pipeline=[dict(type='LoadImageFromFile')] + _pipeline(load_kw, keys + ['gt_extremes'][:0])))
    ds = build_dataset(cfg)
    assert len(ds) == 9 and ds.flag.tolist() == [1, 0, 1] * 3 and os.path.isabs(ds.dataset.ann_file) and ann
    np.random.seed(0)
    loader = build_dataloader(ds, samples_per_gpu=2, workers_per_gpu=workers, dist=False, shuffle=True, seed=5)
    seen = 0
    for batch in loader:
        data = scatter(batch, device='cpu')
        img = data['img']
        assert img.dim() == 4 and img.shape[0] == 2 and img.shape[2] % 32 == 0 and img.shape[3] % 32 == 0
        shapes = {m['ori_shape'][1] > m['ori_shape'][0] for m in data['img_metas']}
        assert len(shapes) == 1                                             # one aspect-ratio group per mini-batch
        for b, lab, m in zip(data['gt_bboxes'], data['gt_labels'], data['gt_masks']):
            assert b.shape[0] == lab.shape[0] == len(m) and b.dtype == torch.float32 and lab.dtype == torch.int64
        seen += 1
    assert seen == len(loader) == 5                                         # groups of 6 and 3 images -> 3 + 2 (padded) batches
    dist_loader = build_dataloader(ds, 2, 0, dist=True, shuffle=True, rank=1, world_size=2)
    assert len(dist_loader) == 3
    test_ds = build_dataset(dict(type='CocoDataset', ann_file=ann, img_prefix=str(tmp_path), test_mode=True, pipeline=[
        dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=[(200, 120), (100, 60)], flip=True, transforms=[
            dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), dict(type='Normalize', **NORM),
            dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]))
    item = test_ds[0]
    assert len(item['img']) == 4 and [m.data['flip'] for m in item['img_metas']] == [False, True, False, True]
    assert torch.equal(item['img'][1][:, :, :item['img_metas'][1].data['img_shape'][1]].flip(-1)[:, :5, :5],
                       item['img'][0][:, :5, :5])


@pytest.mark.skipif(not os.path.isdir('/root/reference/code'), reason='the reference tree is not on this machine')
def test_single_process_group_sampler_equals_reference():
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    from oracle.ref_harness import bootstrap
    bootstrap.load_reference()
    from mmdet.datasets.samplers import GroupSampler as Ref
    for n, spg in ((23, 2), (40, 4), (7, 3)):
        ds = _dataset(n, n)
        np.random.seed(n)
        ours = list(GroupSampler(ds, spg))
        np.random.seed(n)
        assert ours == list(Ref(ds, spg)) and len(ours) % spg == 0


# (training from COCO files end to end, with validation: tests/test_evaluation.py::test_train_detector_with_validation)


def test_extreme_points_equal_reference_fixture_and_tool(tmp_path):
    from lsnet_amd.data.extreme_points import add_extreme_points, extreme_points
    gold = np.load(os.path.join(GOLD, 'data_pipeline.npz'))
    n = int(gold['extreme/n'])
    assert n > 40
    for i in range(n):
        got, ref = extreme_points(gold[f'extreme/{i}/in']), gold[f'extreme/{i}/out']
        assert got.dtype == ref.dtype and np.array_equal(got, ref), i
    with open(TINY) as f:
        coco = json.load(f)
    for a in coco['annotations']:
        a.pop('extreme_points')
    src, dst = tmp_path / 'in.json', tmp_path / 'out.json'
    src.write_text(json.dumps(coco))
    assert add_extreme_points(str(src), str(dst)) == len(coco['annotations'])
    out = json.loads(dst.read_text())
    for a in out['annotations']:
        ex = np.array(a['extreme_points'])
        assert ex.shape == (10,)
        x, y, w, h = a['bbox']
        assert ex[8] == x + w / 2 and ex[9] == y + h / 2
        if a['iscrowd']:                                     # run-length mask: runs 37 | 120 | rest, column-major
            hh = a['segmentation']['size'][0]
            cols = np.arange(37, 157) // hh
            assert ex[2] == cols.min() and ex[6] == cols.max()
        else:
            pts = np.array([v for c in a['segmentation'] for v in c]).reshape(-1, 2)
            assert ex[1] == pts[:, 1].min() and ex[5] == pts[:, 1].max() and ex[2] == pts[:, 0].min() and ex[6] == pts[:, 0].max()
    ds = CocoDataset(str(dst), pipeline=[])                  # and the dataset reads what the tool wrote
    assert ds.get_ann_info(0)['extremes'].shape[1] == 10


def test_cpv_semantic_targets_equal_reference_fixture():
    from lsnet_amd.data.pipelines import PIPELINES
    from lsnet_amd.utils.registry import build_from_cfg
    gold = np.load(os.path.join(GOLD, 'data_pipeline.npz'))
    stage = build_from_cfg(dict(type='LoadRPDV2Annotations', num_classes=5), PIPELINES)
    res = stage(dict(gt_bboxes=gold['rpdv2/boxes'], gt_labels=gold['rpdv2/labels'], pad_shape=(96, 128, 3)))
    assert np.array_equal(res['gt_sem_map'], gold['rpdv2/sem']) and res['gt_sem_map'].sum() > 0
    assert np.array_equal(res['gt_sem_weights'], gold['rpdv2/weights'])
    bundle = build_from_cfg(dict(type='RPDV2FormatBundle'), PIPELINES)(dict(res, img=np.zeros((96, 128, 3), np.float32)))
    assert bundle['gt_sem_map'].stack and bundle['gt_sem_map'].data.shape == (5, 12, 16) and bundle['img'].data.shape == (3, 96, 128)


@pytest.mark.skipif(not os.path.isdir('/root/reference/code/cocoapi'), reason='the reference tree is not on this machine')
@pytest.mark.parametrize('test_mode', [False, True])
def test_datasets_equal_reference_datasets(test_mode):
    """The reference's own CocoDataset / CocoPoseDataset over its vendored COCO api (oracle/ref_harness) list the same
    images in the same order with the same group flags, categories and parsed annotations."""
    import contextlib
    import io
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    from oracle.ref_harness import bootstrap
    bootstrap.load_reference()
    from mmdet.datasets import CocoDataset as RefCoco
    from mmdet.datasets import CocoPoseDataset as RefPose
    for ours_cls, ref_cls in ((CocoDataset, RefCoco), (CocoPoseDataset, RefPose)):
        with contextlib.redirect_stdout(io.StringIO()):
            ref = ref_cls(TINY, pipeline=[], test_mode=test_mode)
        ours = ours_cls(TINY, pipeline=[], test_mode=test_mode)
        assert [d['id'] for d in ours.data_infos] == [d['id'] for d in ref.data_infos]
        assert ours.cat_ids == ref.cat_ids and ours.cat2label == ref.cat2label and ours.img_ids == ref.img_ids
        if not test_mode:
            assert np.array_equal(ours.flag, ref.flag)
        for idx in range(len(ours)):
            a, b = ours.get_ann_info(idx), ref.get_ann_info(idx)
            assert sorted(a) == sorted(b)
            for k in a:
                if isinstance(a[k], np.ndarray):
                    assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (idx, k)
                else:
                    assert a[k] == b[k], (idx, k)
            assert ours.get_cat_ids(idx) == ref.get_cat_ids(idx)


def test_native_image_ops_equal_numpy_formulation(monkeypatch):
    """liblsnet_host.so (csrc/host/image.cpp) and the numpy statement of the same arithmetic in data/geometry.py return
    identical arrays: 8-bit fixed-point and float32 bilinear resize, normalisation with / without channel reversal."""
    rng = np.random.RandomState(0)
    cases = []
    for i in range(30):
        sh, sw, dh, dw = int(rng.randint(2, 90)), int(rng.randint(2, 90)), int(rng.randint(1, 200)), int(rng.randint(1, 200))
        img = rng.randint(0, 256, (sh, sw) + [(), (1,), (3,), (4,)][i % 4]).astype(np.uint8)
        cases.append((img, (dw, dh)))
    cases.append((rng.randint(0, 256, (480, 640, 3)).astype(np.uint8), (1067, 800)))
    native = [(G.imresize(im, sz), G.imresize(im.astype(np.float32), sz)) for im, sz in cases]
    mean, std = np.array(NORM['mean'], np.float32), np.array(NORM['std'], np.float32)
    big = cases[-1][0]
    native_norm = [G.imnormalize(big, mean, std, True), G.imnormalize(big, mean, std, False),
                   G.imnormalize(big.astype(np.float32), mean, std, True)]
    assert G._native() is not None, 'liblsnet_host.so did not load'
    monkeypatch.setenv('LSNET_NUMPY_IMAGE', '1')
    assert G._native() is None
    for (im, sz), (a, af) in zip(cases, native):
        b, bf = G.imresize(im, sz), G.imresize(im.astype(np.float32), sz)
        assert a.dtype == b.dtype == np.uint8 and a.shape == b.shape and np.array_equal(a, b), (im.shape, sz)
        assert af.dtype == bf.dtype == np.float32 and np.array_equal(af, bf), (im.shape, sz)
    ref_norm = [G.imnormalize(big, mean, std, True), G.imnormalize(big, mean, std, False),
                G.imnormalize(big.astype(np.float32), mean, std, True)]
    for a, b in zip(native_norm, ref_norm):
        assert a.dtype == b.dtype and a.flags['C_CONTIGUOUS'] and np.array_equal(a, b)


def test_deferred_pipeline_equals_eager_pipeline(tmp_path):
    """`LoadImageFromFile(defer_to_device=True)`: the stages only record what they would do to the pixels; `scatter`
    then builds the batch (host library on a CPU target) -- same tensors, same ground truth, same meta data as the eager
    pipeline, flips and batch padding included."""
    ann = _write_images(str(tmp_path))
    cls, load_kw, keys = TASKS['segm']

    def make(defer):
        return build_dataset(dict(type='CocoDataset', ann_file=ann, img_prefix=str(tmp_path), pipeline=[
            dict(type='LoadImageFromFile', defer_to_device=defer)] + _pipeline(load_kw, keys, scale=(333, 200))))
    eager, deferred = make(False), make(True)
    flips = set()
    for seed in (0, 1, 2):
        runs = []
        for ds in (eager, deferred):
            np.random.seed(seed)
            runs.append([scatter(b, 'cpu') for b in build_dataloader(ds, 2, 0, dist=False, shuffle=True, seed=seed)])
        assert len(runs[0]) == len(runs[1]) == 2
        for x, y in zip(*runs):
            assert x['img'].dtype == y['img'].dtype == torch.float32 and torch.equal(x['img'], y['img'])
            assert all(torch.equal(p, q) for p, q in zip(x['gt_bboxes'], y['gt_bboxes']))
            for m, n in zip(x['img_metas'], y['img_metas']):
                assert m['flip'] == n['flip'] and tuple(m['img_shape']) == tuple(n['img_shape'])
                assert tuple(m['pad_shape']) == tuple(n['pad_shape']) and np.array_equal(m['scale_factor'], n['scale_factor'])
                flips.add(m['flip'])
            for p, q in zip(x['gt_masks'], y['gt_masks']):
                assert all(np.array_equal(a, b) for oa, ob in zip(p.masks, q.masks) for a, b in zip(oa, ob))
    assert flips == {False, True}
