"""Image preparation on the device the batch is going to (the MI355X-native input path).

The reference resizes, mirrors, normalises and pads every image on CPU workers and ships float32 tensors
(mmdet/datasets/pipelines/transforms.py: Resize / RandomFlip / Normalize / Pad, mmcv/parallel/collate.py).  With
`LoadImageFromFile(defer_to_device=True)` those stages only do their bookkeeping -- sizes, scale factors, flips of the
ground truth -- and the decoded 8-bit image travels as it is; `prepare_batch` then makes the network input where it is
needed: on a GPU one `lsn_image_prep_u8` launch per image writes straight into the channels-last float batch
(csrc/image.hip; the upload is 1 byte per sample of the ORIGINAL image instead of 4 bytes per sample of the padded
one), on the CPU the host library does the same arithmetic.  Both give the tensor the eager pipeline gives, bit for bit
(tests/test_data_pipeline.py, tests/test_zz_device_prep_gpu.py)."""
import ctypes

import numpy as np
import torch

from . import geometry as G


def plan_of(meta):
    """What has to happen to the pixels, from the meta data the (deferred) pipeline stages recorded."""
    cfg = meta.get('img_norm_cfg') or dict(mean=np.zeros(3, np.float32), std=np.ones(3, np.float32), to_rgb=False)
    h, w = meta['img_shape'][:2]
    return dict(size=(int(w), int(h)), pad=tuple(int(v) for v in meta['pad_shape'][:2]), flip=bool(meta.get('flip', False)),
                direction=meta.get('flip_direction', 'horizontal'), mean=np.asarray(cfg['mean'], np.float32),
                std=np.asarray(cfg['std'], np.float32), to_rgb=bool(cfg['to_rgb']))


def _host_image(img, plan):
    out = G.imresize(img, plan['size'])
    if plan['flip']:
        out = G.imflip(out, plan['direction'])
    out = G.imnormalize(out, plan['mean'], plan['std'], plan['to_rgb'])
    return G.impad(out, plan['pad'], 0)


def prepare_batch(images, metas, device=None):
    """images: list of HxWxC uint8 arrays / tensors (decoded, BGR); metas: their meta dicts.
    -> (B, C, H, W) float32 tensor on `device`, channels-last, H x W = the largest padded shape of the batch."""
    plans = [plan_of(m) for m in metas]
    device = torch.device(device if device is not None else 'cpu')
    B, c = len(images), int(images[0].shape[2])
    H, W = max(p['pad'][0] for p in plans), max(p['pad'][1] for p in plans)
    batch = torch.empty((B, c, H, W), dtype=torch.float32, device=device).contiguous(memory_format=torch.channels_last)
    if device.type != 'cuda':
        view = batch.permute(0, 2, 3, 1)                                   # (B, H, W, C) view of the same memory
        view.zero_()
        for b, (img, p) in enumerate(zip(images, plans)):
            arr = img.numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
            out = _host_image(arr, p)
            view[b, :out.shape[0], :out.shape[1]] = torch.from_numpy(out)
        return batch
    from .. import _lib
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    f32p = ctypes.POINTER(ctypes.c_float)
    keep = []
    for b, (img, p) in enumerate(zip(images, plans)):
        t = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
        t = t.contiguous()
        if t.device.type == 'cpu':
            t = (t if t.is_pinned() else t.pin_memory()).to(device, non_blocking=True)
        keep.append(t)
        mean = np.ascontiguousarray(p['mean'], dtype=np.float32)
        inv = np.ascontiguousarray((1.0 / p['std'].astype(np.float64)).astype(np.float32))
        horizontal = p['flip'] and p['direction'] == 'horizontal'
        vertical = p['flip'] and p['direction'] == 'vertical'
        _lib.check(lib.lsn_image_prep_u8(ctypes.c_void_p(t.data_ptr()), int(t.shape[0]), int(t.shape[1]), c, p['size'][1],
                                         p['size'][0], int(horizontal), int(vertical), mean.ctypes.data_as(f32p),
                                         inv.ctypes.data_as(f32p), int(p['to_rgb']), ctypes.c_float(0.0),
                                         ctypes.c_void_p(batch[b].data_ptr()), H, W, stream))
    return batch
