"""Run-length instance masks: the interface of `pycocotools.mask` (cocoapi/pycocotools/pycocotools/mask.py:78-110,
_mask.pyx:96-308) over liblsnet_host.so (include/lsnet_host.h).  Same objects as COCO's api -- an RLE is
{'size': [h, w], 'counts': bytes} in its compressed ASCII form -- so result files are interchangeable.

Used by the segm task's evaluation: predicted 36-vertex polygons -> RLE (`frPyObjects` + `merge`,
mmdet/core/mask/utils.py:65-68), ground-truth polygons / crowd regions -> RLE (coco.py:486-505), IoU matrices
(cocoeval.py:126-152)."""
import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc', 'liblsnet_host.so')
_lib = None
_u32p, _f64p, _szp, _u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_size_t), C.POINTER(C.c_uint8)

EXPORTS = ('lsn_rle_from_polygon', 'lsn_rle_from_bbox', 'lsn_rle_merge', 'lsn_rle_area', 'lsn_rle_to_bbox',
           'lsn_rle_iou', 'lsn_bbox_iou', 'lsn_rle_encode', 'lsn_rle_decode', 'lsn_rle_to_string', 'lsn_rle_from_string',
           'lsn_coco_match', 'lsn_image_resize_bilinear_u8', 'lsn_image_resize_bilinear_f32', 'lsn_image_normalize_u8',
           'lsn_image_normalize_f32', 'lsn_soft_nms', 'lsn_nms_match', 'lsn_nms_host_f32', 'lsn_nms_host_f64')


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f'{_SO} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"`')
        L = C.CDLL(_SO)
        L.lsn_rle_from_polygon.restype = C.c_size_t
        L.lsn_rle_from_polygon.argtypes = [_f64p, C.c_size_t, C.c_uint32, C.c_uint32, _u32p, C.c_size_t]
        L.lsn_rle_from_bbox.restype = C.c_size_t
        L.lsn_rle_from_bbox.argtypes = [_f64p, C.c_uint32, C.c_uint32, _u32p, C.c_size_t]
        L.lsn_rle_merge.restype = C.c_size_t
        L.lsn_rle_merge.argtypes = [_u32p, _szp, C.c_size_t, C.c_int, _u32p, C.c_size_t]
        L.lsn_rle_area.restype = None
        L.lsn_rle_area.argtypes = [_u32p, _szp, C.c_size_t, _u32p]
        L.lsn_rle_to_bbox.restype = None
        L.lsn_rle_to_bbox.argtypes = [_u32p, _szp, C.c_size_t, _u32p, _u32p, _f64p]
        L.lsn_rle_iou.restype = None
        L.lsn_rle_iou.argtypes = [_u32p, _szp, _u32p, _u32p, C.c_size_t, _u32p, _szp, _u32p, _u32p, C.c_size_t, _u8p, _f64p]
        L.lsn_bbox_iou.restype = None
        L.lsn_bbox_iou.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _u8p, _f64p]
        L.lsn_rle_encode.restype = C.c_size_t
        L.lsn_rle_encode.argtypes = [_u8p, C.c_uint32, C.c_uint32, _u32p, C.c_size_t]
        L.lsn_rle_decode.restype = None
        L.lsn_rle_decode.argtypes = [_u32p, C.c_size_t, _u8p, C.c_size_t]
        L.lsn_rle_to_string.restype = C.c_size_t
        L.lsn_rle_to_string.argtypes = [_u32p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.lsn_rle_from_string.restype = C.c_size_t
        L.lsn_rle_from_string.argtypes = [C.c_char_p, _u32p, C.c_size_t]
        L.lsn_coco_match.restype = None
        L.lsn_coco_match.argtypes = [_f64p, C.c_size_t, C.c_size_t, _u8p, _u8p, _f64p, C.c_size_t, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64)]
        f32p = C.POINTER(C.c_float)
        L.lsn_image_resize_bilinear_u8.restype = C.c_int
        L.lsn_image_resize_bilinear_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.lsn_image_resize_bilinear_f32.restype = C.c_int
        L.lsn_image_resize_bilinear_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int]
        L.lsn_image_normalize_u8.restype = None
        L.lsn_image_normalize_u8.argtypes = [_u8p, C.c_size_t, C.c_int, f32p, f32p, C.c_int, f32p]
        L.lsn_image_normalize_f32.restype = None
        L.lsn_image_normalize_f32.argtypes = [f32p, C.c_size_t, C.c_int, f32p, f32p, C.c_int, f32p]
        L.lsn_soft_nms.restype = C.c_size_t
        L.lsn_soft_nms.argtypes = [f32p, C.c_size_t, C.c_float, C.c_int, C.c_float, C.c_float, f32p]
        L.lsn_nms_match.restype = C.c_size_t
        L.lsn_nms_match.argtypes = [f32p, C.POINTER(C.c_int64), C.c_size_t, C.c_float, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.lsn_nms_host_f32.restype = C.c_size_t
        L.lsn_nms_host_f32.argtypes = [f32p, C.POINTER(C.c_int64), C.c_size_t, C.c_float, C.POINTER(C.c_int64)]
        L.lsn_nms_host_f64.restype = C.c_size_t
        L.lsn_nms_host_f64.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_size_t, C.c_float, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _grow(call, cap=4096):
    """Run `call(buffer, cap) -> m` until the buffer is large enough; returns the m runs."""
    while True:
        buf = np.empty(cap, dtype=np.uint32)
        m = call(_p(buf, _u32p), cap)
        if m <= cap:
            return buf[:m].copy()
        cap = m


def counts_to_string(counts):
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    cap = 6 * len(counts) + 1
    s = C.create_string_buffer(cap)
    n = lib().lsn_rle_to_string(_p(counts, _u32p), len(counts), s, cap)
    return s.raw[:n]


def string_to_counts(s):
    s = s.encode('ascii') if isinstance(s, str) else bytes(s)
    buf = np.empty(len(s) + 1, dtype=np.uint32)
    m = lib().lsn_rle_from_string(s, _p(buf, _u32p), len(buf))
    return buf[:m].copy()


def _counts_of(rle):
    c = rle['counts']
    return np.ascontiguousarray(c, dtype=np.uint32) if isinstance(c, (list, tuple, np.ndarray)) else string_to_counts(c)


def _obj(counts, h, w):
    return {'size': [int(h), int(w)], 'counts': counts_to_string(counts)}


def _flat(rles):
    parts = [_counts_of(r) for r in rles]
    offsets = np.zeros(len(parts) + 1, dtype=np.uintp)
    offsets[1:] = np.cumsum([len(p) for p in parts])
    flat = np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros(0), dtype=np.uint32)
    hs = np.ascontiguousarray([r['size'][0] for r in rles], dtype=np.uint32)
    ws = np.ascontiguousarray([r['size'][1] for r in rles], dtype=np.uint32)
    return flat, offsets, hs, ws


def frPoly(polys, h, w):
    out = []
    for p in polys:
        xy = np.ascontiguousarray(p, dtype=np.double)
        k = int(len(p) / 2)
        out.append(_obj(_grow(lambda b, c: lib().lsn_rle_from_polygon(_p(xy, _f64p), k, h, w, b, c)), h, w))
    return out


def frBbox(bb, h, w):
    bb = np.ascontiguousarray(np.asarray(bb, dtype=np.double).reshape(-1, 4))
    return [_obj(_grow(lambda b, c, row=row: lib().lsn_rle_from_bbox(_p(row, _f64p), h, w, b, c)), h, w)
            for row in (np.ascontiguousarray(r) for r in bb)]


def frUncompressedRLE(objs, h, w):
    return [_obj(np.asarray(o['counts'], dtype=np.uint32), o['size'][0], o['size'][1]) for o in objs]


def frPyObjects(pyobj, h, w):
    """Polygons / boxes / uncompressed RLE(s) -> RLE(s), with the dispatch rules of _mask.pyx:288-308."""
    if type(pyobj) == np.ndarray:
        return frBbox(pyobj, h, w)
    if type(pyobj) == list and len(pyobj[0]) == 4:
        return frBbox(pyobj, h, w)
    if type(pyobj) == list and len(pyobj[0]) > 4:
        return frPoly(pyobj, h, w)
    if type(pyobj) == list and type(pyobj[0]) == dict and 'counts' in pyobj[0] and 'size' in pyobj[0]:
        return frUncompressedRLE(pyobj, h, w)
    if type(pyobj) == list and len(pyobj) == 4:
        return frBbox([pyobj], h, w)[0]
    if type(pyobj) == list and len(pyobj) > 4:
        return frPoly([pyobj], h, w)[0]
    if type(pyobj) == dict and 'counts' in pyobj and 'size' in pyobj:
        return frUncompressedRLE([pyobj], h, w)[0]
    raise Exception('input type is not supported.')


def merge(rles, intersect=0):
    flat, offsets, hs, ws = _flat(rles)
    if len(rles) == 0:
        return _obj(np.zeros(0, np.uint32), 0, 0)
    if len(rles) > 1 and (len(set(hs.tolist())) > 1 or len(set(ws.tolist())) > 1):
        return _obj(np.zeros(0, np.uint32), 0, 0)            # masks of different grids: empty result, as the reference
    n = len(rles)
    out = _grow(lambda b, c: lib().lsn_rle_merge(_p(flat, _u32p), _p(offsets, _szp), n, int(intersect), b, c),
                cap=max(4096, len(flat) + 1))
    return _obj(out, hs[0], ws[0])


def area(rles):
    single = not isinstance(rles, list)
    flat, offsets, _, _ = _flat([rles] if single else rles)
    out = np.zeros(len(offsets) - 1, dtype=np.uint32)
    lib().lsn_rle_area(_p(flat, _u32p), _p(offsets, _szp), len(out), _p(out, _u32p))
    return out[0] if single else out


def toBbox(rles):
    single = not isinstance(rles, list)
    flat, offsets, hs, ws = _flat([rles] if single else rles)
    out = np.zeros((len(hs), 4), dtype=np.double)
    lib().lsn_rle_to_bbox(_p(flat, _u32p), _p(offsets, _szp), len(hs), _p(hs, _u32p), _p(ws, _u32p), _p(out, _f64p))
    return out[0] if single else out


def encode(mask):
    """(h, w) or (h, w, n) uint8 -> RLE or list of RLEs"""
    single = mask.ndim == 2
    m3 = mask.reshape(mask.shape[0], mask.shape[1], -1)
    h, w = m3.shape[:2]
    out = []
    for i in range(m3.shape[2]):
        col = np.ascontiguousarray(m3[:, :, i].T.reshape(-1), dtype=np.uint8)        # column-major pixel order
        out.append(_obj(_grow(lambda b, c: lib().lsn_rle_encode(_p(col, _u8p), h, w, b, c)), h, w))
    return out[0] if single else out


def decode(rles):
    single = not isinstance(rles, list)
    lst = [rles] if single else rles
    h, w = lst[0]['size']
    out = np.zeros((h, w, len(lst)), dtype=np.uint8)
    for i, r in enumerate(lst):
        cnt = _counts_of(r)
        buf = np.zeros(h * w, dtype=np.uint8)
        lib().lsn_rle_decode(_p(cnt, _u32p), len(cnt), _p(buf, _u8p), h * w)
        out[:, :, i] = buf.reshape(w, h).T
    return out[:, :, 0] if single else out


def iou(dt, gt, iscrowd):
    """(m, n) IoU of detections x ground truths; both lists of RLEs or both boxes [x, y, w, h] (list or (k,4) array).
    `iscrowd[g]`: the union is replaced by the detection's area.  [] when either side is empty."""
    def kind(o):
        if len(o) == 0:
            return 'empty'
        if isinstance(o, np.ndarray) or all(len(x) == 4 and isinstance(x, (list, np.ndarray)) for x in o):
            return 'box'
        if all(isinstance(x, dict) for x in o):
            return 'rle'
        raise Exception('list input can be bounding box (Nx4) or RLEs ([RLE])')
    kd, kg = kind(dt), kind(gt)
    if 'empty' in (kd, kg):
        return []
    if kd != kg:
        raise Exception('The dt and gt should have the same data type, either RLEs, list or np.ndarray')
    crowd = np.ascontiguousarray(iscrowd, dtype=np.uint8)
    cp = _p(crowd, _u8p) if crowd.size else None
    if kd == 'box':
        d = np.ascontiguousarray(np.asarray(dt, dtype=np.double).reshape(-1, 4))
        g = np.ascontiguousarray(np.asarray(gt, dtype=np.double).reshape(-1, 4))
        out = np.zeros((len(d), len(g)), dtype=np.double)
        lib().lsn_bbox_iou(_p(d, _f64p), len(d), _p(g, _f64p), len(g), cp, _p(out, _f64p))
        return out
    df, do, dh, dw = _flat(dt)
    gf, go, gh, gw = _flat(gt)
    out = np.zeros((len(dt), len(gt)), dtype=np.double)
    lib().lsn_rle_iou(_p(df, _u32p), _p(do, _szp), _p(dh, _u32p), _p(dw, _u32p), len(dt), _p(gf, _u32p), _p(go, _szp),
                      _p(gh, _u32p), _p(gw, _u32p), len(gt), cp, _p(out, _f64p))
    return out
