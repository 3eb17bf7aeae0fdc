"""NMS ops -- mirror of mmdet/ops/nms/nms_wrapper.py:7-59 (`nms`) and :119-157 (`batched_nms`).

The device path never leaves the GPU: IoU bitmask kernel + on-device sweep (lsn_nms).  The keep
indices are bit-identical to the reference's CPU/GPU kernels (same IoU arithmetic, same order).
CPU tensors and ndarrays without a device id take the host library's `nms_cpu` counterpart (lsn_nms_host_f32 / _f64,
the reference's own dispatch: nms_wrapper.py:33-37 -> nms_ext.nms -> nms_cpu for non-CUDA tensors), float32 or float64."""
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
    elif not dets_th.is_cuda:
        inds = _nms_host(dets_th, float(iou_thr))
    else:
        inds = get_backend(dets_th).nms(dets_th.float(), float(iou_thr))
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def _nms_host(dets, iou_thr):
    """nms_cpu (cpu/nms_cpu.cpp:7-71): float32 / float64 boxes on the host, order = torch.sort of the scores."""
    import ctypes as C
    if dets.dtype not in (torch.float32, torch.float64):
        dets = dets.float()
    a = dets.detach().contiguous().numpy().reshape(-1, 5)
    # stable: equal scores keep their index order, as the device kernel and the oracle do (the reference's unstable
    # torch.sort leaves the order of exact ties unspecified)
    order = torch.sort(dets[:, 4], dim=0, descending=True, stable=True)[1].numpy()
    keep = np.zeros(len(a), dtype=np.int64)
    i64 = C.POINTER(C.c_int64)
    if a.dtype == np.float64:
        k = _host().lsn_nms_host_f64(a.ctypes.data_as(C.POINTER(C.c_double)), order.ctypes.data_as(i64), len(a), iou_thr,
                                     keep.ctypes.data_as(i64))
    else:
        k = _host().lsn_nms_host_f32(a.ctypes.data_as(C.POINTER(C.c_float)), order.ctypes.data_as(i64), len(a), iou_thr,
                                     keep.ctypes.data_as(i64))
    return torch.from_numpy(keep[:k].copy())


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
    if nms_type not in ('nms', 'soft_nms'):
        raise NotImplementedError(f'nms type {nms_type!r} is not one of nms_wrapper.py\'s operators')
    op = nms if nms_type == 'nms' else soft_nms            # soft_nms runs on the host, as in the reference
    dets, keep = op(torch.cat([boxes_for_nms, scores[:, None]], -1), **cfg)
    return torch.cat([bboxes[keep], dets[:, -1:]], -1), keep


def _host():
    from ..evaluation.mask import lib
    return lib()


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """Soft NMS on the host (nms_wrapper.py:62-116; the reference has no device version either): returns
    (new_dets (k, 5), inds (k,)) in the type of the input."""
    import ctypes as C
    if isinstance(dets, torch.Tensor):
        is_tensor, arr = True, dets.detach().cpu().numpy()
    elif isinstance(dets, np.ndarray):
        is_tensor, arr = False, dets
    else:
        raise TypeError(f'dets must be either a Tensor or numpy array, but got {type(dets)}')
    codes = {'linear': 1, 'gaussian': 2}
    if method not in codes:
        raise ValueError(f'Invalid method for SoftNMS: {method}')
    a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 5)
    out = np.zeros((len(a), 6), dtype=np.float32)
    f32 = C.POINTER(C.c_float)
    k = _host().lsn_soft_nms(a.ctypes.data_as(f32), len(a), float(iou_thr), codes[method], float(sigma), float(min_score),
                             out.ctypes.data_as(f32)) if len(a) else 0
    new_dets, inds = out[:k, :5], out[:k, 5].astype(np.int64)
    if is_tensor:
        return torch.from_numpy(new_dets).to(device=dets.device, dtype=dets.dtype), torch.from_numpy(inds).to(dets.device)
    return new_dets.astype(arr.dtype), inds


def nms_match(dets, thresh):
    """Groups of boxes that NMS would merge (nms_wrapper.py:160-191): one index array per kept box, the keeper first."""
    import ctypes as C
    if dets.shape[0] == 0:
        return []
    assert dets.shape[-1] == 5, f'inputs dets.shape should be (N, 5), but get {dets.shape}'
    is_tensor = isinstance(dets, torch.Tensor)
    t = dets.detach().cpu().float() if is_tensor else torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32))
    order = t[:, 4].sort(0, descending=True)[1].contiguous().numpy()
    a = np.ascontiguousarray(t.numpy())
    flat, start = np.zeros(len(a), np.int64), np.zeros(len(a) + 1, np.int64)
    i64 = C.POINTER(C.c_int64)
    g = _host().lsn_nms_match(a.ctypes.data_as(C.POINTER(C.c_float)), order.ctypes.data_as(i64), len(a), float(thresh),
                              flat.ctypes.data_as(i64), start.ctypes.data_as(i64))
    groups = [flat[start[k]:start[k + 1]].copy() for k in range(g)]
    return [dets.new_tensor(m, dtype=torch.long) for m in groups] if is_tensor else groups
