from .coco_index import CocoIndex, rle_decode
from .datasets import (DATASETS, CocoDataset, CocoPoseDataset, ConcatDataset, CustomDataset, RepeatDataset,
                       build_dataloader, build_dataset)
from .gt_formats import flip_extremes, flip_keypoints, flip_polygons, polygon_landmarks, resample_polygon
from .masks import PolygonMasks
from .pipelines import PIPELINES, Compose
from .samplers import DistributedGroupSampler, GroupSampler
from .synthetic import synthetic_batch

__all__ = ['synthetic_batch', 'DistributedGroupSampler', 'GroupSampler', 'resample_polygon', 'polygon_landmarks',
           'flip_extremes', 'flip_polygons', 'flip_keypoints', 'PolygonMasks', 'PIPELINES', 'Compose', 'DATASETS',
           'CustomDataset', 'CocoDataset', 'CocoPoseDataset', 'ConcatDataset', 'RepeatDataset', 'build_dataset',
           'build_dataloader', 'CocoIndex', 'rle_decode']
