"""Test loops (mmdet/apis/test.py:13-207): run the detector over a test loader, bring every result into the form the
dataset's `evaluate` takes, gather the parts of all ranks.

Per image the LSNet detectors return [per-class boxes, per-class landmark vectors]:
  bbox task  -> the boxes only (the extreme points are not scored by COCO);
  segm task  -> (boxes, per-class lists of RLE masks rasterised from the 36-vertex polygons at the ORIGINAL image size);
  pose tasks -> unchanged (boxes + keypoints, scored as COCO keypoints);
  LSCPVDetector returns per-class boxes already."""
import pickle

import numpy as np
import torch
import torch.distributed as dist

from ..evaluation import mask as mask_util
from ..parallel.data_container import scatter


def encode_poly_results(poly_results, img_h, img_w):
    """per-class (n, 2m) polygon arrays -> per-class lists of RLEs (mmdet/core/mask/utils.py:65-85)"""
    return [[mask_util.merge(mask_util.frPyObjects([np.asarray(p).copy().tolist()], img_h, img_w)) for p in cls]
            for cls in poly_results]


def _finish(result, head, metas):
    task = getattr(head, 'task', None)
    if type(head).__name__ == 'LSHead':
        if task == 'bbox':
            return result[0]
        if task == 'segm':
            h, w = metas[0]['ori_shape'][:2]
            return result[0], encode_poly_results(result[1], h, w)
    return result


def _module(model):
    return model.module if hasattr(model, 'module') else model


def _run(model, data):
    m = _module(model)
    dev = next(m.parameters()).device
    data = scatter(data, dev, channels_last=dev.type == 'cuda')
    with torch.no_grad():
        result = model(return_loss=False, rescale=True, **data)
    return _finish(result, m.bbox_head, data['img_metas'][0])


def single_gpu_test(model, data_loader, **kwargs):
    model.eval()
    return [_run(model, data) for data in data_loader]


def collect_results(part, size):
    """Every rank holds the results of samples rank, rank + world, ... (DistributedSampler without shuffle): rank 0
    gets them interleaved back into dataset order, cut to `size` (the sampler pads)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return part[:size]
    world = dist.get_world_size()
    parts = [None] * world
    dist.all_gather_object(parts, pickle.dumps(part))
    if dist.get_rank() != 0:
        return None
    parts = [pickle.loads(p) for p in parts]
    ordered = []
    for group in zip(*parts):
        ordered.extend(group)
    return ordered[:size]


def multi_gpu_test(model, data_loader, tmpdir=None, gpu_collect=False, **kwargs):
    model.eval()
    part = [_run(model, data) for data in data_loader]
    return collect_results(part, len(data_loader.dataset))
