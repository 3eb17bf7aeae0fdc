"""The per-image processing pipeline named by `train_pipeline` / `test_pipeline` of the reference's configs
(configs/_base_/datasets/coco_lsvr.py:5-33, coco_lsvr_segm.py, coco_lsvr_pose.py): a list of `dict(type=...)` stages
acting on one `results` dict.  Same stage names, arguments, result keys and random draws (`np.random`) as
mmdet/datasets/pipelines/{compose,loading,transforms,formating,test_time_aug}.py, so the config files load unchanged
and a fixed numpy seed gives the same flips / scales.

Ground-truth layouts: `gt_bboxes` (G,4) xyxy, `gt_extremes` (G,10) = 4 extreme points +
centre, `gt_keypoints` (G,51) = 17 x (x,y,v), `gt_masks` = PolygonMasks of 36-vertex clockwise contours that start
at the vertex nearest the top-centre of their bounding box."""
import os.path as osp
import warnings
from collections.abc import Sequence

import numpy as np
import torch

from ..parallel.data_container import DataContainer as DC
from ..utils.registry import Registry, build_from_cfg
from . import geometry as G
from .gt_formats import polygon_landmarks
from .masks import PolygonMasks

PIPELINES = Registry('pipeline')


@PIPELINES.register_module()
class Compose:
    """compose.py:8-51: stages run in order; a stage returning None drops the sample."""

    def __init__(self, transforms):
        assert isinstance(transforms, Sequence)
        self.transforms = []
        for t in transforms:
            if isinstance(t, dict):
                self.transforms.append(build_from_cfg(t, PIPELINES))
            elif callable(t):
                self.transforms.append(t)
            else:
                raise TypeError('transform must be callable or a dict')

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data

    def __repr__(self):
        return type(self).__name__ + '(' + ''.join(f'\n    {t}' for t in self.transforms) + '\n)'


@PIPELINES.register_module()
class LoadImageFromFile:
    """loading.py:12-77."""

    def __init__(self, to_float32=False, color_type='color', file_client_args=None, defer_to_device=False):
        """`defer_to_device`: keep the decoded 8-bit image untouched through Resize / RandomFlip / Normalize / Pad (they
        only record what they would do) and let `data.device_prep.prepare_batch` make the network input on the device."""
        self.to_float32, self.color_type, self.defer_to_device = to_float32, color_type, defer_to_device
        if file_client_args not in (None, dict(backend='disk')):
            raise ValueError('only the disk backend exists here')

    def __call__(self, results):
        name = results['img_info']['filename']
        filename = osp.join(results['img_prefix'], name) if results.get('img_prefix') is not None else name
        img = G.imread(filename, self.color_type)
        if self.to_float32:
            img = img.astype(np.float32)
        results.update(filename=filename, ori_filename=name, img=img, img_shape=img.shape, ori_shape=img.shape,
                       img_fields=['img'])
        if self.defer_to_device:
            assert img.dtype == np.uint8 and img.ndim == 3
            results['img_deferred'] = True
        return results

    def __repr__(self):
        return f"{type(self).__name__}(to_float32={self.to_float32}, color_type='{self.color_type}')"


@PIPELINES.register_module()
class LoadAnnotations:
    """loading.py:158-520.  `with_mask` needs `poly2mask=False`: instance masks stay polygons, resampled to
    `num_contour_points` landmark vertices (`unify_polygons`, :422-441 -> gt_formats.polygon_landmarks)."""

    def __init__(self, with_bbox=True, with_label=True, with_mask=False, with_seg=False, with_extreme=False,
                 with_keypoint=False, poly2mask=True, file_client_args=None, spline_num=10, num_contour_points=128):
        self.with_bbox, self.with_label, self.with_mask, self.with_seg = with_bbox, with_label, with_mask, with_seg
        self.with_extreme, self.with_keypoint, self.poly2mask = with_extreme, with_keypoint, poly2mask
        self.spline_num, self.num_points = spline_num, num_contour_points
        if with_mask and poly2mask:
            raise NotImplementedError('bitmap instance masks are not on the LSNet path; set poly2mask=False')
        if with_seg:
            raise NotImplementedError('semantic segmentation maps are not on the LSNet path')

    def __call__(self, results):
        ann = results['ann_info']
        if self.with_bbox:
            results['gt_bboxes'] = ann['bboxes'].copy()
            ignore = ann.get('bboxes_ignore', None)
            if ignore is not None:
                results['gt_bboxes_ignore'] = ignore.copy()
                results['bbox_fields'].append('gt_bboxes_ignore')
            results['bbox_fields'].append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = ann['labels'].copy()
        if self.with_mask:
            h, w = results['img_info']['height'], results['img_info']['width']
            polys = [polygon_landmarks(p, ann['bboxes'][i], self.num_points, self.spline_num)
                     for i, p in enumerate(ann['masks'])]
            results['gt_masks'] = PolygonMasks(polys, h, w)
            results['mask_fields'].append('gt_masks')
        if self.with_extreme:
            results['gt_extremes'] = ann['extremes'].copy()
            results['extreme_fields'].append('gt_extremes')
        if self.with_keypoint:
            results['gt_keypoints'] = ann['keypoints'].copy()
            results['keypoint_fields'].append('gt_keypoints')
        return results

    def __repr__(self):
        return (f'{type(self).__name__}(with_bbox={self.with_bbox}, with_extreme={self.with_extreme}, '
                f'with_keypoint={self.with_keypoint}, with_label={self.with_label}, with_mask={self.with_mask}, '
                f'poly2mask={self.poly2mask})')


@PIPELINES.register_module()
class Resize:
    """transforms.py:24-301.  Scale selection: one scale, a `ratio_range` around it, or several scales sampled by
    value / range; `keep_ratio` fits the image inside (long edge, short edge)."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True):
        if img_scale is None:
            self.img_scale = None
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            assert all(isinstance(s, tuple) for s in self.img_scale)
        if ratio_range is not None:
            assert len(self.img_scale) == 1
        else:
            assert multiscale_mode in ('value', 'range')
        self.multiscale_mode, self.ratio_range, self.keep_ratio = multiscale_mode, ratio_range, keep_ratio

    @staticmethod
    def random_select(img_scales):
        idx = np.random.randint(len(img_scales))
        return img_scales[idx], idx

    @staticmethod
    def random_sample(img_scales):
        assert len(img_scales) == 2
        longs, shorts = [max(s) for s in img_scales], [min(s) for s in img_scales]
        long_edge = np.random.randint(min(longs), max(longs) + 1)
        short_edge = np.random.randint(min(shorts), max(shorts) + 1)
        return (long_edge, short_edge), None

    @staticmethod
    def random_sample_ratio(img_scale, ratio_range):
        lo, hi = ratio_range
        assert isinstance(img_scale, tuple) and len(img_scale) == 2 and lo <= hi
        ratio = np.random.random_sample() * (hi - lo) + lo
        return (int(img_scale[0] * ratio), int(img_scale[1] * ratio)), None

    def _random_scale(self, results):
        if self.ratio_range is not None:
            scale, idx = self.random_sample_ratio(self.img_scale[0], self.ratio_range)
        elif len(self.img_scale) == 1:
            scale, idx = self.img_scale[0], 0
        elif self.multiscale_mode == 'range':
            scale, idx = self.random_sample(self.img_scale)
        else:
            scale, idx = self.random_select(self.img_scale)
        results['scale'], results['scale_idx'] = scale, idx

    def __call__(self, results):
        if 'scale' not in results:
            if 'scale_factor' in results:
                factor = results['scale_factor']
                assert isinstance(factor, float)
                results['scale'] = tuple(int(x * factor) for x in results['img'].shape[:2])[::-1]
            else:
                self._random_scale(results)
        else:
            assert 'scale_factor' not in results, 'scale and scale_factor cannot be both set.'
        for key in results.get('img_fields', ['img']):
            h, w = results[key].shape[:2]
            if results.get('img_deferred'):                    # sizes only; the pixels are resized on the device
                new_w, new_h = G.rescale_size((w, h), results['scale']) if self.keep_ratio else results['scale']
                shape = (int(new_h), int(new_w)) + tuple(results[key].shape[2:])
                w_scale, h_scale = new_w / w, new_h / h
                results['img_shape'] = results['pad_shape'] = shape
                results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
                results['keep_ratio'] = self.keep_ratio
                continue
            if self.keep_ratio:
                img = G.imrescale(results[key], results['scale'])
                w_scale, h_scale = img.shape[1] / w, img.shape[0] / h
            else:
                img, w_scale, h_scale = G.imresize(results[key], results['scale'], return_scale=True)
            results[key] = img
            results['img_shape'] = results['pad_shape'] = img.shape
            results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
            results['keep_ratio'] = self.keep_ratio
        ih, iw = results['img_shape'][:2]
        sf = results['scale_factor']
        for key in results.get('bbox_fields', []):
            b = results[key] * sf
            b[:, 0::2] = np.clip(b[:, 0::2], 0, iw)
            b[:, 1::2] = np.clip(b[:, 1::2], 0, ih)
            results[key] = b
        for key in results.get('extreme_fields', []):
            e = results[key] * np.tile(sf[:2], (1, 5))
            e[:, 0::2] = np.clip(e[:, 0::2], 0, iw)
            e[:, 1::2] = np.clip(e[:, 1::2], 0, ih)
            results[key] = e
        for key in results.get('keypoint_fields', []):           # in place, as the reference does
            k = results[key]
            k[:, 0::3] = np.clip(k[:, 0::3] * sf[0], 0, iw)
            k[:, 1::3] = np.clip(k[:, 1::3] * sf[1], 0, ih)
        for key in results.get('mask_fields', []):
            if results[key] is None:
                continue
            results[key] = (results[key].rescale(results['scale']) if self.keep_ratio
                            else results[key].resize(results['img_shape'][:2]))
        return results

    def __repr__(self):
        return (f'{type(self).__name__}(img_scale={self.img_scale}, multiscale_mode={self.multiscale_mode}, '
                f'ratio_range={self.ratio_range}, keep_ratio={self.keep_ratio})')


KEYPOINT_FLIP_PAIRS = ((1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:304-459.  Boxes swap their two x (or y) sides; extreme points mirror and the left/right (or
    top/bottom) points trade places; keypoints mirror and left/right joints swap (horizontal only, as in the
    reference); contours mirror and, with `keep_poly_clockwise`, are re-ordered to stay clockwise."""

    def __init__(self, flip_ratio=None, direction='horizontal', keep_poly_clockwise=True):
        if flip_ratio is not None:
            assert 0 <= flip_ratio <= 1
        assert direction in ('horizontal', 'vertical')
        self.flip_ratio, self.direction, self.keep_poly_clockwise = flip_ratio, direction, keep_poly_clockwise

    @staticmethod
    def bbox_flip(bboxes, img_shape, direction):
        assert bboxes.shape[-1] % 4 == 0
        out = bboxes.copy()
        if direction == 'horizontal':
            w = img_shape[1]
            out[..., 0::4], out[..., 2::4] = w - bboxes[..., 2::4], w - bboxes[..., 0::4]
        elif direction == 'vertical':
            h = img_shape[0]
            out[..., 1::4], out[..., 3::4] = h - bboxes[..., 3::4], h - bboxes[..., 1::4]
        else:
            raise ValueError(f"Invalid flipping direction '{direction}'")
        return out

    @staticmethod
    def extreme_flip(extremes, img_shape, direction):
        """(…, 10k): [top, left, bottom, right, centre] x (x, y)."""
        assert extremes.shape[-1] % 10 == 0
        e = extremes.reshape(extremes.shape[:-1] + (extremes.shape[-1] // 10, 5, 2))
        out = e.copy()
        if direction == 'horizontal':
            out[..., 0] = img_shape[1] - e[..., 0]
            out[..., [1, 3], :] = out[..., [3, 1], :]
        elif direction == 'vertical':
            out[..., 1] = img_shape[0] - e[..., 1]
            out[..., [0, 2], :] = out[..., [2, 0], :]
        else:
            raise ValueError(f"Invalid flipping direction '{direction}'")
        return out.reshape(extremes.shape)

    @staticmethod
    def keypoint_flip(keypoints, img_shape, direction):
        assert keypoints.shape[-1] % 17 == 0
        out = keypoints.copy()
        if direction == 'horizontal':
            out[:, 0::3] = img_shape[1] - out[:, 0::3]
            k = out.reshape(out.shape[0], out.shape[1] // 3, 3)        # (also for zero instances)
            perm = np.arange(k.shape[1])
            for a, b in KEYPOINT_FLIP_PAIRS:
                perm[a], perm[b] = b, a
            out = k[:, perm].reshape(keypoints.shape)
        elif direction == 'vertical':
            out[:, 1::3] = img_shape[0] - out[:, 1::3]
        else:
            raise ValueError(f"Invalid flipping direction '{direction}'")
        return out

    def __call__(self, results):
        if 'flip' not in results:
            results['flip'] = bool(np.random.rand() < self.flip_ratio)
        if 'flip_direction' not in results:
            results['flip_direction'] = self.direction
        if results['flip']:
            d, shape = results['flip_direction'], results['img_shape']
            for key in results.get('img_fields', ['img']):
                if not results.get('img_deferred'):
                    results[key] = G.imflip(results[key], d)
            for key in results.get('bbox_fields', []):
                results[key] = self.bbox_flip(results[key], shape, d)
            for key in results.get('extreme_fields', []):
                results[key] = self.extreme_flip(results[key], shape, d)
            for key in results.get('keypoint_fields', []):
                results[key] = self.keypoint_flip(results[key], shape, d)
            for key in results.get('mask_fields', []):
                results[key] = results[key].flip(d, self.keep_poly_clockwise)
        return results

    def __repr__(self):
        return f'{type(self).__name__}(flip_ratio={self.flip_ratio})'


@PIPELINES.register_module()
class Normalize:
    """transforms.py:532-571."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb

    def __call__(self, results):
        for key in results.get('img_fields', ['img']):
            if not results.get('img_deferred'):
                results[key] = G.imnormalize(results[key], self.mean, self.std, self.to_rgb)
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results

    def __repr__(self):
        return f'{type(self).__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})'


@PIPELINES.register_module()
class Pad:
    """transforms.py:462-529: pad bottom/right to a fixed size or to a multiple of `size_divisor`."""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert (size is None) != (size_divisor is None)
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val

    def __call__(self, results):
        if results.get('img_deferred'):
            assert self.pad_val == 0, 'the device path pads with zeros'
            h, w = results['img_shape'][:2]
            if self.size is not None:
                ph, pw = self.size
            else:
                ph, pw = (int(np.ceil(v / self.size_divisor)) * self.size_divisor for v in (h, w))
            results['pad_shape'] = (ph, pw) + tuple(results['img_shape'][2:])
        else:
            for key in results.get('img_fields', ['img']):
                padded = (G.impad(results[key], self.size, self.pad_val) if self.size is not None
                          else G.impad_to_multiple(results[key], self.size_divisor, self.pad_val))
                results[key] = padded
            results['pad_shape'] = padded.shape
        results['pad_fixed_size'], results['pad_size_divisor'] = self.size, self.size_divisor
        for key in results.get('mask_fields', []):
            results[key] = results[key].pad(results['pad_shape'][:2], pad_val=self.pad_val)
        return results

    def __repr__(self):
        return f'{type(self).__name__}(size={self.size}, size_divisor={self.size_divisor}, pad_val={self.pad_val})'


def to_tensor(data):
    """formating.py:12-34."""
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data)
    if isinstance(data, Sequence) and not isinstance(data, str):
        return torch.tensor(data)
    if isinstance(data, int):
        return torch.LongTensor([data])
    if isinstance(data, float):
        return torch.FloatTensor([data])
    raise TypeError(f'type {type(data)} cannot be converted to tensor.')


def _chw(img):
    if img.ndim < 3:
        img = np.expand_dims(img, -1)
    return np.ascontiguousarray(img.transpose(2, 0, 1))


@PIPELINES.register_module()
class ToTensor:

    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for k in self.keys:
            results[k] = to_tensor(results[k])
        return results

    def __repr__(self):
        return f'{type(self).__name__}(keys={self.keys})'


@PIPELINES.register_module()
class ImageToTensor:
    """formating.py:64-95: HWC -> CHW tensor."""

    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for k in self.keys:
            results[k] = to_tensor(_chw(results[k]))
        return results

    def __repr__(self):
        return f'{type(self).__name__}(keys={self.keys})'


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formating.py:175-248: image -> CHW tensor in a stacking container; ragged ground truth -> tensors in plain
    containers; polygon masks stay on the host."""

    TENSOR_KEYS = ('proposals', 'gt_bboxes', 'gt_bboxes_ignore', 'gt_labels', 'gt_extremes', 'gt_keypoints')

    def __call__(self, results):
        if 'img' in results:
            img = results['img']
            results.setdefault('pad_shape', img.shape)
            results.setdefault('scale_factor', 1.0)
            nch = 1 if img.ndim < 3 else img.shape[2]
            results.setdefault('img_norm_cfg', dict(mean=np.zeros(nch, dtype=np.float32),
                                                    std=np.ones(nch, dtype=np.float32), to_rgb=False))
            if results.get('img_deferred'):        # decoded HxWxC bytes; batched and prepared by prepare_batch
                results['img'] = DC(to_tensor(np.ascontiguousarray(img)), stack=False)
            else:
                results['img'] = DC(to_tensor(_chw(img)), stack=True)
        for key in self.TENSOR_KEYS:
            if key in results:
                results[key] = DC(to_tensor(results[key]))
        if 'gt_masks' in results:
            results['gt_masks'] = DC(results['gt_masks'], cpu_only=True)
        return results

    def __repr__(self):
        return type(self).__name__


@PIPELINES.register_module()
class Collect:
    """formating.py:251-325: keep `keys`, gather `meta_keys` into `img_metas` (host-only container)."""

    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape',
                                        'scale_factor', 'flip', 'flip_direction', 'img_norm_cfg')):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {'img_metas': DC({k: results[k] for k in self.meta_keys}, cpu_only=True)}
        for k in self.keys:
            data[k] = results[k]
        return data

    def __repr__(self):
        return f'{type(self).__name__}(keys={self.keys}, meta_keys={self.meta_keys})'


@PIPELINES.register_module()
class MultiScaleFlipAug:
    """test_time_aug.py:9-117: run `transforms` once per (scale, flip, direction); values become lists."""

    def __init__(self, transforms, img_scale=None, scale_factor=None, flip=False, flip_direction='horizontal'):
        self.transforms = Compose(transforms)
        assert (img_scale is None) != (scale_factor is None), 'Must have but only one variable can be setted'
        if img_scale is not None:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            self.scale_key = 'scale'
            assert all(isinstance(s, tuple) for s in self.img_scale)
        else:
            self.img_scale = scale_factor if isinstance(scale_factor, list) else [scale_factor]
            self.scale_key = 'scale_factor'
        self.flip = flip
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]
        if not self.flip and self.flip_direction != ['horizontal']:
            warnings.warn('flip_direction has no effect when flip is set to False')
        if self.flip and not any(t['type'] == 'RandomFlip' for t in transforms):
            warnings.warn('flip has no effect when RandomFlip is not in transforms')

    def __call__(self, results):
        aug = []
        for scale in self.img_scale:
            for flip in ([False, True] if self.flip else [False]):
                for direction in self.flip_direction:
                    r = results.copy()
                    r[self.scale_key], r['flip'], r['flip_direction'] = scale, flip, direction
                    aug.append(self.transforms(r))
        return {k: [d[k] for d in aug] for k in aug[0]}

    def __repr__(self):
        return (f'{type(self).__name__}(transforms={self.transforms}, img_scale={self.img_scale}, flip={self.flip}, '
                f'flip_direction={self.flip_direction})')


@PIPELINES.register_module()
class LoadRPDV2Annotations:
    """Box-level semantic targets of the corner-point-verification head at stride 8 (loading_reppointsv2.py:8-60):
    per class, 1 inside every ground-truth box (cells int(x1/8)..int(x2/8) inclusive) and a weight of 1/box area;
    boxes are painted from the largest to the smallest so that small objects win overlaps.  Runs after `Pad`."""

    def __init__(self, num_classes=80):
        self.num_classes = num_classes

    def __call__(self, results):
        boxes, labels = results['gt_bboxes'], results['gt_labels']
        h, w = int(results['pad_shape'][0] / 8), int(results['pad_shape'][1] / 8)
        areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        sem = np.zeros((self.num_classes, h, w), dtype=np.float32)
        weights = np.zeros((self.num_classes, h, w), dtype=np.float32)
        for i in np.argsort(areas)[::-1]:
            x1, y1, x2, y2 = (int(v / 8) for v in boxes[i])
            sem[labels[i], y1:y2 + 1, x1:x2 + 1] = 1
            weights[labels[i], y1:y2 + 1, x1:x2 + 1] = 1 / areas[i]
        results['gt_sem_map'], results['gt_sem_weights'] = sem, weights
        return results

    def __repr__(self):
        return f'{type(self).__name__}(num_classes={self.num_classes})'


@PIPELINES.register_module()
class RPDV2FormatBundle(DefaultFormatBundle):
    """formating_reppointsv2.py:11-96: the default bundle (without keypoints) plus the stacked semantic maps."""

    TENSOR_KEYS = ('proposals', 'gt_bboxes', 'gt_bboxes_ignore', 'gt_labels', 'gt_extremes')

    def __call__(self, results):
        results = super().__call__(results)
        for key in ('gt_sem_map', 'gt_sem_weights'):
            if key in results:
                results[key] = DC(to_tensor(results[key]), stack=True)
        if 'gt_contours' in results:
            results['gt_contours'] = DC(to_tensor(results['gt_contours']))
        return results
