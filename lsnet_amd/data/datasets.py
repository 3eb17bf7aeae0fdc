"""Datasets of the LSNet configs and the loader builder (mmdet/datasets/{custom,coco,coco_pose,dataset_wrappers,
builder}.py): COCO json (+ the `extreme_points` field written by tools/gen_extreme_points.py) -> per-image
`ann_info` -> pipeline -> collated batches, one `DataLoader` per process / GPU.

`CocoDataset`     -- 80 classes, boxes + extreme points + polygon segmentations (bbox and segm tasks);
`CocoPoseDataset` -- 'person' only, boxes + 17x3 keypoints (pose tasks).
Annotation filtering is the reference's: `ignore` flag, boxes with no overlap with the image, area <= 0 or sides
< 1 px, unknown categories dropped; crowd boxes go to `bboxes_ignore` (coco.py:120-185, coco_pose.py:110-172)."""
import os.path as osp
import random
from functools import partial

import numpy as np
from torch.utils.data import ConcatDataset as _ConcatDataset
from torch.utils.data import DataLoader, Dataset, DistributedSampler

from ..parallel.data_container import collate
from ..utils.registry import Registry, build_from_cfg
from .coco_index import CocoIndex
from .pipelines import Compose
from .samplers import DistributedGroupSampler, GroupSampler

DATASETS = Registry('dataset')

COCO_CLASSES = (
    'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light',
    'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant',
    'bear', 'zebra', 'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard',
    'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
    'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange', 'broccoli',
    'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet',
    'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone', 'microwave', 'oven', 'toaster', 'sink',
    'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush')


@DATASETS.register_module()
class CustomDataset(Dataset):
    """custom.py:13-321.  Subclasses provide `load_annotations` / `get_ann_info`."""

    CLASSES = None

    def __init__(self, ann_file, pipeline, classes=None, data_root=None, img_prefix='', seg_prefix=None,
                 proposal_file=None, test_mode=False, filter_empty_gt=True):
        self.ann_file, self.data_root, self.img_prefix, self.seg_prefix = ann_file, data_root, img_prefix, seg_prefix
        self.proposal_file, self.test_mode, self.filter_empty_gt = proposal_file, test_mode, filter_empty_gt
        self.CLASSES = self.get_classes(classes)
        if data_root is not None:
            def rooted(p):
                return p if (p is None or osp.isabs(p)) else osp.join(data_root, p)
            self.ann_file = osp.join(data_root, ann_file) if not osp.isabs(ann_file) else ann_file
            self.img_prefix, self.seg_prefix = rooted(self.img_prefix), rooted(self.seg_prefix)
        if proposal_file is not None:
            raise NotImplementedError('proposal files belong to two-stage detectors, not to the LSNet path')
        self.proposals = None
        self.data_infos = self.load_annotations(self.ann_file)
        if self.custom_classes:
            self.data_infos = self.get_subset_by_classes()
        if not test_mode:
            self.data_infos = [self.data_infos[i] for i in self._filter_imgs()]
            self._set_group_flag()
        self.pipeline = Compose(pipeline)

    def __len__(self):
        return len(self.data_infos)

    def load_annotations(self, ann_file):
        raise NotImplementedError

    def get_ann_info(self, idx):
        return self.data_infos[idx]['ann']

    def get_cat_ids(self, idx):
        return self.data_infos[idx]['ann']['labels'].astype(np.int64).tolist()

    def pre_pipeline(self, results):
        results.update(img_prefix=self.img_prefix, seg_prefix=self.seg_prefix, proposal_file=self.proposal_file,
                       bbox_fields=[], extreme_fields=[], mask_fields=[], seg_fields=[], keypoint_fields=[])

    def _filter_imgs(self, min_size=32):
        return [i for i, info in enumerate(self.data_infos) if min(info['width'], info['height']) >= min_size]

    def _set_group_flag(self):
        """group 1 = wider than tall; a mini-batch only mixes images of one group (less padding)."""
        self.flag = np.zeros(len(self), dtype=np.uint8)
        for i, info in enumerate(self.data_infos):
            if info['width'] / info['height'] > 1:
                self.flag[i] = 1

    def _rand_another(self, idx):
        return np.random.choice(np.where(self.flag == self.flag[idx])[0])

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is None:
                idx = self._rand_another(idx)
                continue
            return data

    def prepare_train_img(self, idx):
        results = dict(img_info=self.data_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    def prepare_test_img(self, idx):
        results = dict(img_info=self.data_infos[idx])
        self.pre_pipeline(results)
        return self.pipeline(results)

    @classmethod
    def get_classes(cls, classes=None):
        if classes is None:
            cls.custom_classes = False
            return cls.CLASSES
        cls.custom_classes = True
        if isinstance(classes, str):
            with open(classes) as f:
                return [line.rstrip('\n') for line in f]
        if isinstance(classes, (tuple, list)):
            return classes
        raise ValueError(f'Unsupported type {type(classes)} of classes.')

    def get_subset_by_classes(self):
        return self.data_infos


@DATASETS.register_module()
class CocoDataset(CustomDataset):

    CLASSES = COCO_CLASSES
    INSTANCE_FIELD = ('extremes', 'extreme_points', 10)        # ann_info key, json key, row width

    def load_annotations(self, ann_file):
        self.coco = CocoIndex(ann_file)
        self.cat_ids = self.coco.get_cat_ids(cat_names=self.CLASSES)
        self.cat2label = {cat_id: i for i, cat_id in enumerate(self.cat_ids)}
        self.img_ids = self.coco.get_img_ids()
        infos = []
        for i in self.img_ids:
            info = self.coco.load_imgs([i])[0]
            info['filename'] = info['file_name']
            infos.append(info)
        return infos

    def _anns_of(self, idx):
        return self.coco.load_anns(self.coco.get_ann_ids(img_ids=[self.data_infos[idx]['id']]))

    def get_ann_info(self, idx):
        return self._parse_ann_info(self.data_infos[idx], self._anns_of(idx))

    def get_cat_ids(self, idx):
        return [a['category_id'] for a in self._anns_of(idx)]

    def _filter_imgs(self, min_size=32):
        with_ann = set(a['image_id'] for a in self.coco.anns.values())
        keep = []
        for i, info in enumerate(self.data_infos):
            if self.filter_empty_gt and self.img_ids[i] not in with_ann:
                continue
            if min(info['width'], info['height']) >= min_size:
                keep.append(i)
        return keep

    def get_subset_by_classes(self):
        ids = set()
        for cat_id in self.cat_ids:
            ids |= set(self.coco.cat_img_map[cat_id])
        self.img_ids = list(ids)
        infos = []
        for i in self.img_ids:
            info = self.coco.load_imgs([i])[0]
            info['filename'] = info['file_name']
            infos.append(info)
        return infos

    def _parse_ann_info(self, img_info, ann_info):
        key, json_key, width = self.INSTANCE_FIELD
        boxes, labels, ignored, masks, extra = [], [], [], [], []
        for ann in ann_info:
            if ann.get('ignore', False):
                continue
            x1, y1, w, h = ann['bbox']
            inter_w = max(0, min(x1 + w, img_info['width']) - max(x1, 0))
            inter_h = max(0, min(y1 + h, img_info['height']) - max(y1, 0))
            if inter_w * inter_h == 0 or ann['area'] <= 0 or w < 1 or h < 1:
                continue
            if ann['category_id'] not in self.cat_ids:
                continue
            box = [x1, y1, x1 + w, y1 + h]
            if ann.get('iscrowd', False):
                ignored.append(box)
            else:
                boxes.append(box)
                labels.append(self.cat2label[ann['category_id']])
                masks.append(ann['segmentation'])
                extra.append(ann[json_key])
        if boxes:
            boxes, labels = np.array(boxes, dtype=np.float32), np.array(labels, dtype=np.int64)
            extra = np.array(extra, dtype=np.float32)
        else:
            boxes, labels = np.zeros((0, 4), dtype=np.float32), np.array([], dtype=np.int64)
            extra = np.zeros((0, width), dtype=np.float32)
        ignored = np.array(ignored, dtype=np.float32) if ignored else np.zeros((0, 4), dtype=np.float32)
        return {'bboxes': boxes, 'labels': labels, 'bboxes_ignore': ignored, 'masks': masks, key: extra,
                'seg_map': img_info['filename'].replace('jpg', 'png')}

    # ---- detections -> COCO result records -> metrics (coco.py:187-507) -----------------------------------
    @staticmethod
    def xyxy2xywh(bbox):
        b = bbox.tolist()
        return [b[0], b[1], b[2] - b[0], b[3] - b[1]]

    def _records(self, results, pick_boxes, extra=None):
        out = []
        for idx in range(len(self)):
            img_id = self.img_ids[idx]
            per_class = pick_boxes(results[idx])
            for label, dets in enumerate(per_class):
                for i in range(dets.shape[0]):
                    rec = dict(image_id=img_id, bbox=self.xyxy2xywh(dets[i]), score=float(dets[i][4]),
                               category_id=self.cat_ids[label])
                    if extra is not None:
                        rec.update(extra(results[idx], label, i))
                    out.append(rec)
        return out

    def _det2json(self, results):
        """`results[i]` = per-class list of (n, 5) arrays `[x1, y1, x2, y2, score]` of image i."""
        return self._records(results, lambda r: r)

    def _segm2json(self, results):
        """`results[i]` = (per-class boxes, per-class lists of RLE dicts)."""
        def seg(r, label, i):
            rle = dict(r[1][label][i])
            if isinstance(rle['counts'], bytes):
                rle['counts'] = rle['counts'].decode()
            return dict(segmentation=rle)
        return self._records(results, lambda r: r[0]), self._records(results, lambda r: r[0], seg)

    def results2json(self, results, outfile_prefix):
        import json
        files = {}

        def dump(records, kind, suffix):
            files[kind] = f'{outfile_prefix}.{suffix}.json'
            with open(files[kind], 'w') as f:
                json.dump(records, f)
        if isinstance(results[0], list):
            dump(self._det2json(results), 'bbox', 'bbox')
            files['proposal'] = files['bbox']
        elif isinstance(results[0], tuple):
            boxes, segms = self._segm2json(results)
            dump(boxes, 'bbox', 'bbox')
            files['proposal'] = files['bbox']
            dump(segms, 'segm', 'segm')
        else:
            raise TypeError('invalid type of results')
        return files

    def format_results(self, results, jsonfile_prefix=None, **kwargs):
        import tempfile
        assert isinstance(results, list), 'results must be a list'
        assert len(results) == len(self), f'The length of results is not equal to the dataset len: {len(results)} != {len(self)}'
        tmp_dir = None
        if jsonfile_prefix is None:
            tmp_dir = tempfile.TemporaryDirectory()
            jsonfile_prefix = osp.join(tmp_dir.name, 'results')
        return self.results2json(results, jsonfile_prefix), tmp_dir

    ALLOWED_METRICS = ('bbox', 'segm')

    def evaluate(self, results, metric='bbox', logger=None, jsonfile_prefix=None, classwise=False, **kwargs):
        """COCO AP / AR of `results` (one entry per image, test-mode order) -> {'bbox_mAP': ..., 'bbox_mAP_50': ...}.
        Metrics: 'bbox', 'segm' (and 'keypoints' on CocoPoseDataset); the proposal metrics of two-stage detectors are
        not on the LSNet path."""
        from ..evaluation.coco_eval import CocoEval, load_results
        log = logger if callable(logger) else (logger.info if hasattr(logger, 'info') else (lambda s: None))
        metrics = metric if isinstance(metric, list) else [metric]
        for m in metrics:
            if m not in self.ALLOWED_METRICS:
                raise KeyError(f'metric {m} is not supported')
        files, tmp_dir = self.format_results(results, jsonfile_prefix)
        out = {}
        for m in metrics:
            log(f'Evaluating {m}...')
            if m not in files:
                raise KeyError(f'{m} is not in results')
            try:
                dt = load_results(self.coco, files[m])
            except IndexError:
                log('The testing results of the whole dataset is empty.')
                break
            ev = CocoEval(self.coco, dt, m)
            ev.params.cat_ids, ev.params.img_ids = self.cat_ids, self.img_ids
            ev.evaluate()
            ev.accumulate()
            ev.summarize(printer=log)
            if classwise:
                prec = ev.eval['precision']
                assert len(self.cat_ids) == prec.shape[2]
                per_class = {}
                for k, cat_id in enumerate(self.cat_ids):
                    pk = prec[:, :, k, 0, -1]
                    pk = pk[pk > -1]
                    per_class[self.coco.load_cats([cat_id])[0]['name']] = float(np.mean(pk)) if pk.size else float('nan')
                out[f'{m}_classwise_AP'] = per_class
            for i, item in enumerate(('mAP', 'mAP_50', 'mAP_75', 'mAP_s', 'mAP_m', 'mAP_l')):
                out[f'{m}_{item}'] = float(f'{ev.stats[i]:.3f}')
            out[f'{m}_mAP_copypaste'] = ' '.join(f'{v:.3f}' for v in ev.stats[:6])
        if tmp_dir is not None:
            tmp_dir.cleanup()
        return out


@DATASETS.register_module()
class CocoPoseDataset(CocoDataset):
    """coco_pose.py:19-172.  There CLASSES is the bare string 'person'; the COCO api takes a string for a sequence and
    keeps the categories whose name is a SUBSTRING of it -- in COCO files that is the person category alone."""

    CLASSES = ('person',)
    INSTANCE_FIELD = ('keypoints', 'keypoints', 51)
    ALLOWED_METRICS = ('bbox', 'segm', 'keypoints')

    def _det2json(self, results):
        """`results[i]` = [per-class boxes, per-class (n, 34) keypoint arrays] (coco_pose.py:209-224)."""
        return self._records(results, lambda r: r[0])

    def _kps2json(self, results):
        """coco_pose.py:226-247: every keypoint is reported as labelled and visible (v = 1)."""
        def kps(r, label, i):
            k = np.concatenate([r[1][label][i].reshape(-1, 2), np.ones((17, 1), dtype=np.float32)], axis=1)
            return dict(keypoints=k.reshape(51).tolist())
        return self._records(results, lambda r: r[0], kps)

    def results2json(self, results, outfile_prefix):
        import json
        if not (isinstance(results[0], list) and len(results[0]) == 2 and isinstance(results[0][0], list)):
            return super().results2json(results, outfile_prefix)
        files = {'bbox': f'{outfile_prefix}.bbox.json', 'keypoints': f'{outfile_prefix}.kps.json'}
        files['proposal'] = files['bbox']
        for kind, records in (('bbox', self._det2json(results)), ('keypoints', self._kps2json(results))):
            with open(files[kind], 'w') as f:
                json.dump(records, f)
        return files


@DATASETS.register_module()
class ConcatDataset(_ConcatDataset):
    """dataset_wrappers.py:8-39."""

    def __init__(self, datasets):
        super().__init__(datasets)
        self.CLASSES = datasets[0].CLASSES
        if hasattr(datasets[0], 'flag'):
            self.flag = np.concatenate([d.flag for d in datasets])

    def get_cat_ids(self, idx):
        if idx < 0:
            if -idx > len(self):
                raise ValueError('absolute value of index should not exceed dataset length')
            idx = len(self) + idx
        d = int(np.searchsorted(self.cumulative_sizes, idx, side='right'))
        return self.datasets[d].get_cat_ids(idx if d == 0 else idx - self.cumulative_sizes[d - 1])


@DATASETS.register_module()
class RepeatDataset:
    """dataset_wrappers.py:42-71."""

    def __init__(self, dataset, times):
        self.dataset, self.times, self.CLASSES = dataset, times, dataset.CLASSES
        if hasattr(dataset, 'flag'):
            self.flag = np.tile(dataset.flag, times)
        self._ori_len = len(dataset)

    def __getitem__(self, idx):
        return self.dataset[idx % self._ori_len]

    def get_cat_ids(self, idx):
        return self.dataset.get_cat_ids(idx % self._ori_len)

    def __len__(self):
        return self.times * self._ori_len


def build_dataset(cfg, default_args=None):
    """builder.py:19-57: list -> ConcatDataset, RepeatDataset wrapper, list of ann_files -> concatenation."""
    if isinstance(cfg, (list, tuple)):
        return ConcatDataset([build_dataset(c, default_args) for c in cfg])
    if cfg['type'] == 'RepeatDataset':
        return RepeatDataset(build_dataset(cfg['dataset'], default_args), cfg['times'])
    if isinstance(cfg.get('ann_file'), (list, tuple)):
        parts = []
        for i, ann in enumerate(cfg['ann_file']):
            c = dict(cfg)
            c['ann_file'] = ann
            for k in ('img_prefix', 'seg_prefix', 'proposal_file'):
                if isinstance(cfg.get(k), (list, tuple)):
                    c[k] = cfg[k][i]
            parts.append(build_dataset(c, default_args))
        return ConcatDataset(parts)
    return build_from_cfg(dict(cfg), DATASETS, default_args)


def worker_init_fn(worker_id, num_workers, rank, seed):
    s = num_workers * rank + worker_id + seed
    np.random.seed(s)
    random.seed(s)


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu, num_gpus=1, dist=True, shuffle=True, seed=None,
                     rank=None, world_size=None, **kwargs):
    """builder.py:60-125.  Distributed: this process loads `samples_per_gpu` images per step for its own GPU, indices
    from DistributedGroupSampler (shuffled, one aspect-ratio group per mini-batch, padded so every rank runs the same
    number of steps).  `pin_memory` stays off as in the reference (the loader's pinning thread does not look inside
    DataContainers); `scatter` pins what it uploads through torch's caching host allocator and copies asynchronously."""
    if rank is None or world_size is None:
        import torch.distributed as td
        ok = td.is_available() and td.is_initialized()
        rank, world_size = (td.get_rank(), td.get_world_size()) if ok else (0, 1)
    if dist:
        sampler = (DistributedGroupSampler(dataset, samples_per_gpu, world_size, rank) if shuffle
                   else DistributedSampler(dataset, world_size, rank, shuffle=False))
        batch_size, num_workers = samples_per_gpu, workers_per_gpu
    else:
        sampler = GroupSampler(dataset, samples_per_gpu) if shuffle else None
        batch_size, num_workers = num_gpus * samples_per_gpu, num_gpus * workers_per_gpu
    init_fn = partial(worker_init_fn, num_workers=num_workers, rank=rank, seed=seed) if seed is not None else None
    kwargs.setdefault('pin_memory', False)
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                      collate_fn=partial(collate, samples_per_gpu=samples_per_gpu), worker_init_fn=init_fn, **kwargs)
