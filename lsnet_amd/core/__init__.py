from .assigners import (BBOX_ASSIGNERS, BBOX_SAMPLERS, AssignResult, ATSSAssigner, BboxOverlaps2D,
                        CentroidAssigner, PointHMAssigner, PseudoSampler, SamplingResult, bbox_overlaps, build_assigner,
                        build_sampler, gaussian_radius)
from .misc import images_to_levels, multi_apply, unmap
from .points import PointGenerator
from .post_processing import multiclass_nms, multiclass_nms_lsvr

__all__ = ['BBOX_ASSIGNERS', 'BBOX_SAMPLERS', 'AssignResult', 'ATSSAssigner', 'BboxOverlaps2D',
           'CentroidAssigner', 'PseudoSampler', 'SamplingResult', 'bbox_overlaps', 'build_assigner',
           'build_sampler', 'images_to_levels', 'multi_apply', 'unmap', 'PointGenerator',
           'multiclass_nms_lsvr', 'multiclass_nms', 'PointHMAssigner', 'gaussian_radius']
