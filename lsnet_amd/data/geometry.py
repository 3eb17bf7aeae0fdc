"""Image helpers of the data pipeline: what the reference gets from `mmcv.image` (cv2 underneath), restated over
numpy + PIL because this stack carries neither (mmcv/image/geometric.py:9-140,225-330, photometric.py:8-41,
io.py:140-190).  Images are HxWxC numpy arrays in BGR order, like the reference's.

`imresize(..., 'bilinear')` follows cv2's INTER_LINEAR for 8-bit images -- pixel centres at half-integers, 11-bit
fixed-point weights, the two-pass rounding of `HResizeLinear`/`VResizeLinear` -- so that a checkpoint trained on
cv2-resized images sees the same pixels; float images are interpolated in float32.  (cv2 is absent from this image, so
the 8-bit path is pinned by its own properties in tests/test_data_pipeline.py, not against cv2 itself.)"""
import ctypes as C
import io
import os

import numpy as np


def _native():
    """liblsnet_host.so (csrc/host/image.cpp).  A missing library raises (build it: `__graft_entry__.build()`); the
    numpy statement of the same arithmetic below runs only on request, LSNET_NUMPY_IMAGE=1 -- the tests use that to
    compare the two."""
    if os.environ.get('LSNET_NUMPY_IMAGE') == '1':
        return None
    from ..evaluation.mask import lib
    return lib()


def rescale_size(old_size, scale, return_scale=False):
    """(w, h) scaled by a factor, or as large as fits inside (long edge, short edge) = sorted(scale)."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        factor = scale
    elif isinstance(scale, tuple):
        factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    else:
        raise TypeError(f'Scale must be a number or tuple of int, but got {type(scale)}')
    new_size = (int(w * float(factor) + 0.5), int(h * float(factor) + 0.5))
    return (new_size, factor) if return_scale else new_size


def _linear_taps(dst, src):
    """Source index and float weight of the right-hand tap for every destination index."""
    scale = 1.0 / (dst / src)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    return s, np.minimum(s + 1, src - 1), f


def _resize_linear_native(img, w, h):
    L = _native()
    if L is None or img.dtype not in (np.uint8, np.float32):
        return None
    sh, sw = img.shape[:2]
    src = np.ascontiguousarray(img)
    c = int(src.size // (sh * sw))
    dst = np.empty((h, w) + img.shape[2:], dtype=img.dtype)
    if img.dtype == np.uint8:
        t = C.POINTER(C.c_uint8)
        rc = L.lsn_image_resize_bilinear_u8(src.ctypes.data_as(t), sh, sw, c, dst.ctypes.data_as(t), h, w)
    else:
        t = C.POINTER(C.c_float)
        rc = L.lsn_image_resize_bilinear_f32(src.ctypes.data_as(t), sh, sw, c, dst.ctypes.data_as(t), h, w)
    return dst if rc == 0 else None


def _resize_linear_u8(img, w, h):
    sh, sw = img.shape[:2]
    x0, x1, fx = _linear_taps(w, sw)
    y0, y1, fy = _linear_taps(h, sh)
    ax1 = np.rint(fx * 2048.0).astype(np.int32)
    ax0 = np.rint((1.0 - fx) * 2048.0).astype(np.int32)
    ay1 = np.rint(fy * 2048.0).astype(np.int32)
    ay0 = np.rint((1.0 - fy) * 2048.0).astype(np.int32)
    src = img.reshape(sh, sw, -1).astype(np.int32)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]       # (sh, w, c), scaled by 2^11
    top = (ay0[:, None, None] * (rows[y0] >> 4)) >> 16
    bot = (ay1[:, None, None] * (rows[y1] >> 4)) >> 16
    out = ((top + bot + 2) >> 2).clip(0, 255).astype(np.uint8)
    return out.reshape((h, w) + img.shape[2:])


def _resize_linear_f32(img, w, h):
    sh, sw = img.shape[:2]
    x0, x1, fx = _linear_taps(w, sw)
    y0, y1, fy = _linear_taps(h, sh)
    src = img.reshape(sh, sw, -1).astype(np.float32)
    rows = src[:, x0] * (1.0 - fx)[None, :, None] + src[:, x1] * fx[None, :, None]
    out = rows[y0] * (1.0 - fy)[:, None, None] + rows[y1] * fy[:, None, None]
    return out.reshape((h, w) + img.shape[2:]).astype(img.dtype, copy=False)


def _resize_nearest(img, w, h):
    sh, sw = img.shape[:2]
    xs = np.minimum(np.floor(np.arange(w) * (sw / w)).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(h) * (sh / h)).astype(np.int64), sh - 1)
    return img[ys][:, xs]


def imresize(img, size, return_scale=False, interpolation='bilinear'):
    """`size` = (w, h)."""
    w, h = int(size[0]), int(size[1])
    sh, sw = img.shape[:2]
    if interpolation == 'nearest':
        out = _resize_nearest(img, w, h)
    elif interpolation == 'bilinear':
        if (w, h) == (sw, sh):
            out = img.copy()
        else:
            out = _resize_linear_native(img, w, h)
            if out is None:
                out = _resize_linear_u8(img, w, h) if img.dtype == np.uint8 else _resize_linear_f32(img, w, h)
    else:
        raise ValueError(f'interpolation {interpolation!r} is not on the LSNet path')
    return (out, w / sw, h / sh) if return_scale else out


def imrescale(img, scale, return_scale=False, interpolation='bilinear'):
    h, w = img.shape[:2]
    new_size, factor = rescale_size((w, h), scale, return_scale=True)
    out = imresize(img, new_size, interpolation=interpolation)
    return (out, factor) if return_scale else out


def imflip(img, direction='horizontal'):
    assert direction in ('horizontal', 'vertical')
    return np.flip(img, axis=1 if direction == 'horizontal' else 0)


def impad(img, shape, pad_val=0):
    """Pad bottom/right to `shape` = (h, w)."""
    shape = tuple(shape)
    if len(shape) < img.ndim:
        shape = shape + (img.shape[-1],)
    assert len(shape) == img.ndim and all(s >= i for s, i in zip(shape, img.shape))
    out = np.empty(shape, dtype=img.dtype)
    out[...] = pad_val
    out[:img.shape[0], :img.shape[1], ...] = img
    return out


def impad_to_multiple(img, divisor, pad_val=0):
    h = int(np.ceil(img.shape[0] / divisor)) * divisor
    w = int(np.ceil(img.shape[1] / divisor)) * divisor
    return impad(img, (h, w), pad_val)


def imnormalize(img, mean, std, to_rgb=True):
    """float32 `(img[..., ::-1 if to_rgb] - mean) * (1 / std)`; the reciprocal is formed in double as in the reference."""
    mean32 = np.ascontiguousarray(np.asarray(mean, dtype=np.float64).reshape(-1).astype(np.float32))
    inv32 = np.ascontiguousarray((1.0 / np.asarray(std, dtype=np.float64).reshape(-1)).astype(np.float32))
    L = _native()
    if L is not None and img.ndim == 3 and img.dtype in (np.uint8, np.float32) and img.shape[2] == mean32.size:
        src = np.ascontiguousarray(img)
        dst = np.empty(img.shape, dtype=np.float32)
        f32p = C.POINTER(C.c_float)
        args = (img.shape[0] * img.shape[1], img.shape[2], mean32.ctypes.data_as(f32p), inv32.ctypes.data_as(f32p),
                int(bool(to_rgb)), dst.ctypes.data_as(f32p))
        if img.dtype == np.uint8:
            L.lsn_image_normalize_u8(src.ctypes.data_as(C.POINTER(C.c_uint8)), *args)
        else:
            L.lsn_image_normalize_f32(src.ctypes.data_as(f32p), *args)
        return dst
    out = img.astype(np.float32)
    if to_rgb:
        out = out[..., ::-1]
    return np.ascontiguousarray((out - mean32.reshape(1, -1)) * inv32.reshape(1, -1))


def imfrombytes(content, flag='color'):
    """Decode an encoded image to BGR uint8 (HxWx3), honouring the EXIF orientation as cv2.imdecode does."""
    from PIL import Image, ImageOps
    im = Image.open(io.BytesIO(content))
    if flag == 'unchanged':
        return np.asarray(im)
    if flag == 'grayscale':
        return np.asarray(ImageOps.exif_transpose(im).convert('L'))
    rgb = np.asarray(ImageOps.exif_transpose(im).convert('RGB'))
    return np.ascontiguousarray(rgb[..., ::-1])


def imread(path, flag='color'):
    with open(path, 'rb') as f:
        return imfrombytes(f.read(), flag)
