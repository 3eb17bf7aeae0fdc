from .gt_formats import flip_extremes, flip_keypoints, flip_polygons, polygon_landmarks, resample_polygon
from .samplers import DistributedGroupSampler
from .synthetic import synthetic_batch

__all__ = ['synthetic_batch', 'DistributedGroupSampler', 'resample_polygon', 'polygon_landmarks', 'flip_extremes', 'flip_polygons',
           'flip_keypoints']
