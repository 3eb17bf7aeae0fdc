"""TEST INFRASTRUCTURE: ctypes binding of the REFERENCE's COCO mask API (oracle/_ref/maskapi.so, built by
oracle/build_ref.py from cocoapi/pycocotools/common/maskApi.c) with the Python-level interface of its Cython module
`pycocotools._mask` (cocoapi/pycocotools/pycocotools/_mask.pyx:96-308).  bootstrap.py installs it under that name so
the reference's pure-Python `pycocotools.coco` / `.cocoeval` / `.mask` import and run unchanged.

RLE objects are the API's own: {'size': [h, w], 'counts': bytes}."""
import ctypes as C
import os

import numpy as np

SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '_ref', 'maskapi.so')


class RLE(C.Structure):
    _fields_ = [('h', C.c_ulong), ('w', C.c_ulong), ('m', C.c_ulong), ('cnts', C.POINTER(C.c_uint))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(SO)
        _lib.rleToString.restype = C.c_void_p
        _lib.rleToString.argtypes = [C.POINTER(RLE)]
        _lib.rleFrString.argtypes = [C.POINTER(RLE), C.c_char_p, C.c_ulong, C.c_ulong]
        _lib.rleFrPoly.argtypes = [C.POINTER(RLE), C.POINTER(C.c_double), C.c_ulong, C.c_ulong, C.c_ulong]
        _lib.rleFrBbox.argtypes = [C.POINTER(RLE), C.POINTER(C.c_double), C.c_ulong, C.c_ulong, C.c_ulong]
        _lib.rleInit.argtypes = [C.POINTER(RLE), C.c_ulong, C.c_ulong, C.c_ulong, C.POINTER(C.c_uint)]
        _lib.rleFree.argtypes = [C.POINTER(RLE)]
        _lib.rleMerge.argtypes = [C.POINTER(RLE), C.POINTER(RLE), C.c_ulong, C.c_int]
        _lib.rleArea.argtypes = [C.POINTER(RLE), C.c_ulong, C.POINTER(C.c_uint)]
        _lib.rleIou.argtypes = [C.POINTER(RLE), C.POINTER(RLE), C.c_ulong, C.c_ulong, C.POINTER(C.c_ubyte), C.POINTER(C.c_double)]
        _lib.bbIou.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_ulong, C.c_ulong, C.POINTER(C.c_ubyte), C.POINTER(C.c_double)]
        _lib.rleToBbox.argtypes = [C.POINTER(RLE), C.POINTER(C.c_double), C.c_ulong]
        _lib.rleDecode.argtypes = [C.POINTER(RLE), C.POINTER(C.c_ubyte), C.c_ulong]
        _lib.rleEncode.argtypes = [C.POINTER(RLE), C.POINTER(C.c_ubyte), C.c_ulong, C.c_ulong, C.c_ulong]
        _libc = C.CDLL(None)
        _libc.free.argtypes = [C.c_void_p]
        _lib._free = _libc.free
    return _lib


class _RLEs:
    def __init__(self, n):
        self.n = n
        self.arr = (RLE * max(n, 1))()

    def __del__(self):
        for i in range(self.n):
            if self.arr[i].cnts:
                lib().rleFree(C.byref(self.arr[i]))


def _fr_string(objs):
    rs = _RLEs(len(objs))
    for i, o in enumerate(objs):
        s = o['counts']
        s = s.encode('ascii') if isinstance(s, str) else bytes(s)
        lib().rleFrString(C.byref(rs.arr[i]), s, o['size'][0], o['size'][1])
    return rs


def _to_string(rs):
    out = []
    for i in range(rs.n):
        p = lib().rleToString(C.byref(rs.arr[i]))
        out.append({'size': [int(rs.arr[i].h), int(rs.arr[i].w)], 'counts': C.string_at(p)})
        lib()._free(p)
    return out


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def encode(mask):
    h, w, n = mask.shape
    m = np.asfortranarray(mask.astype(np.uint8))
    rs = _RLEs(n)
    lib().rleEncode(rs.arr, m.ctypes.data_as(C.POINTER(C.c_ubyte)), h, w, n)
    return _to_string(rs)


def decode(objs):
    rs = _fr_string(objs)
    h, w = int(rs.arr[0].h), int(rs.arr[0].w)
    out = np.zeros((h, w, rs.n), dtype=np.uint8, order='F')
    lib().rleDecode(rs.arr, out.ctypes.data_as(C.POINTER(C.c_ubyte)), rs.n)
    return out


def merge(objs, intersect=0):
    rs = _fr_string(objs)
    r = _RLEs(1)
    lib().rleMerge(rs.arr, r.arr, rs.n, int(intersect))
    return _to_string(r)[0]


def area(objs):
    rs = _fr_string(objs)
    a = np.zeros(rs.n, dtype=np.uint32)
    lib().rleArea(rs.arr, rs.n, a.ctypes.data_as(C.POINTER(C.c_uint)))
    return a


def toBbox(objs):
    rs = _fr_string(objs)
    bb = np.zeros(4 * rs.n, dtype=np.double)
    lib().rleToBbox(rs.arr, _dptr(bb), rs.n)
    return bb.reshape(rs.n, 4)


def frBbox(bb, h, w):
    bb = np.ascontiguousarray(np.asarray(bb, dtype=np.double).reshape(-1, 4))
    rs = _RLEs(bb.shape[0])
    lib().rleFrBbox(rs.arr, _dptr(bb), h, w, bb.shape[0])
    return _to_string(rs)


def frPoly(poly, h, w):
    rs = _RLEs(len(poly))
    for i, p in enumerate(poly):
        a = np.ascontiguousarray(np.array(p, dtype=np.double))
        lib().rleFrPoly(C.byref(rs.arr[i]), _dptr(a), int(len(p) / 2), h, w)
    return _to_string(rs)


def frUncompressedRLE(objs, h, w):
    out = []
    for o in objs:
        cnts = np.ascontiguousarray(np.array(o['counts'], dtype=np.uint32))
        rs = _RLEs(1)
        lib().rleInit(rs.arr, o['size'][0], o['size'][1], len(cnts), cnts.ctypes.data_as(C.POINTER(C.c_uint)))
        out.append(_to_string(rs)[0])
    return out


def frPyObjects(pyobj, h, w):
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


def iou(dt, gt, pyiscrowd):
    def prep(objs):
        if len(objs) == 0:
            return objs
        if type(objs) == np.ndarray:
            return np.ascontiguousarray(objs.reshape(-1, 4).astype(np.double))
        isbox = all(len(o) == 4 and type(o) in (list, np.ndarray) for o in objs)
        if isbox:
            return np.ascontiguousarray(np.array(objs, dtype=np.double).reshape(-1, 4))
        if all(type(o) == dict for o in objs):
            return _fr_string(objs)
        raise Exception('list input can be bounding box (Nx4) or RLEs ([RLE])')
    iscrowd = np.ascontiguousarray(np.array(pyiscrowd, dtype=np.uint8))
    dt, gt = prep(dt), prep(gt)
    m = dt.n if isinstance(dt, _RLEs) else len(dt)
    n = gt.n if isinstance(gt, _RLEs) else len(gt)
    if m == 0 or n == 0:
        return []
    out = np.zeros(m * n, dtype=np.double)
    crowd = iscrowd.ctypes.data_as(C.POINTER(C.c_ubyte)) if iscrowd.size else None
    if isinstance(dt, _RLEs):
        lib().rleIou(dt.arr, gt.arr, m, n, crowd, _dptr(out))
    else:
        lib().bbIou(_dptr(dt), _dptr(gt), m, n, crowd, _dptr(out))
    return out.reshape((m, n), order='F')
