"""NMS ops -- mirror of mmdet/ops/nms/nms_wrapper.py:7-59 (`nms`) and :119-157 (`batched_nms`).

The device path never leaves the GPU: IoU bitmask kernel + on-device sweep (lsn_nms).  The keep
indices are bit-identical to the reference's CPU/GPU kernels (same IoU arithmetic, same order)."""
import numpy as np
import torch

from .backend import get_backend


def nms(dets, iou_thr, device_id=None):
    """dets: (n,5) tensor or ndarray [x1,y1,x2,y2,score].  Returns (dets[inds], inds)."""
    if isinstance(dets, torch.Tensor):
        is_numpy, dets_th = False, dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        device = 'cpu' if device_id is None else f'cuda:{device_id}'
        dets_th = torch.from_numpy(dets).to(device)
    else:
        raise TypeError(f'dets must be either a Tensor or numpy array, but got {type(dets)}')
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        inds = get_backend(dets_th).nms(dets_th.float(), float(iou_thr))
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def batched_nms(bboxes, scores, inds, nms_cfg, class_agnostic=False):
    """Per-class NMS through the coordinate-offset trick (nms_wrapper.py:143-150)."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = bboxes
    else:
        max_coordinate = bboxes.max()
        offsets = inds.to(bboxes) * (max_coordinate + 1)
        boxes_for_nms = bboxes + offsets[:, None]
    nms_type = cfg.pop('type', 'nms')
    if nms_type != 'nms':
        raise NotImplementedError(f'nms type {nms_type!r} is outside the LSNet hot path')
    dets, keep = nms(torch.cat([boxes_for_nms, scores[:, None]], -1), **cfg)
    return torch.cat([bboxes[keep], dets[:, -1:]], -1), keep
