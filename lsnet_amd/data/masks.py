"""`PolygonMasks`: per-instance contours as flat (2m,) arrays `[x0,y0,x1,y1,...]`, a list of components per object
(mmdet/core/mask/structures.py:315-560).  Only what the LSNet segm pipeline touches: indexing, rescale/resize, flip
(optionally re-ordering so that a clockwise contour stays clockwise and keeps its first vertex), pad, crop, areas.
Bitmap masks are not on this path (`poly2mask=False` in every `configs/lsnet/*segm*` file)."""
import numpy as np

from .geometry import rescale_size


class PolygonMasks:

    def __init__(self, masks, height, width):
        assert isinstance(masks, list)
        if masks:
            assert isinstance(masks[0], list) and isinstance(masks[0][0], np.ndarray)
        self.masks, self.height, self.width = masks, height, width

    def __len__(self):
        return len(self.masks)

    def __iter__(self):
        return iter(self.masks)

    def __repr__(self):
        return f'{type(self).__name__}(num_masks={len(self.masks)}, height={self.height}, width={self.width})'

    def __getitem__(self, index):
        if isinstance(index, np.ndarray):
            index = index.tolist()
        if isinstance(index, list):
            picked = [self.masks[i] for i in index]
        else:
            try:
                picked = self.masks[index]
            except Exception:
                raise ValueError(f'Unsupported input of type {type(index)} for indexing!')
        if picked and isinstance(picked[0], np.ndarray):
            picked = [picked]
        return PolygonMasks(picked, self.height, self.width)

    def _map(self, fn, height, width):
        return PolygonMasks([[fn(p.copy()) for p in obj] for obj in self.masks], height, width)

    def resize(self, out_shape, interpolation=None):
        hs, ws = out_shape[0] / self.height, out_shape[1] / self.width

        def f(p):
            p[0::2] *= ws
            p[1::2] *= hs
            return p
        return self._map(f, out_shape[0], out_shape[1])

    def rescale(self, scale, interpolation=None):
        new_w, new_h = rescale_size((self.width, self.height), scale)
        return self.resize((new_h, new_w))

    def flip(self, flip_direction='horizontal', keep_cw=False):
        assert flip_direction in ('horizontal', 'vertical')
        dim, idx = (self.width, 0) if flip_direction == 'horizontal' else (self.height, 1)

        def f(p):
            p[idx::2] = dim - p[idx::2]
            if keep_cw:                                   # reverse the order, first vertex stays first
                q = p.reshape(-1, 2)
                p = np.concatenate([q[:1], q[:0:-1]], 0).reshape(-1)
            return p
        return self._map(f, self.height, self.width)

    def pad(self, out_shape, pad_val=0):
        return PolygonMasks(self.masks, out_shape[0], out_shape[1])

    def crop(self, bbox):
        assert isinstance(bbox, np.ndarray) and bbox.ndim == 1
        bbox = bbox.copy()
        bbox[0::2] = np.clip(bbox[0::2], 0, self.width)
        bbox[1::2] = np.clip(bbox[1::2], 0, self.height)
        x1, y1, x2, y2 = bbox

        def f(p):
            p[0::2] -= x1
            p[1::2] -= y1
            return p
        return self._map(f, np.maximum(y2 - y1, 1), np.maximum(x2 - x1, 1))

    def to_ndarray(self):
        """(N, h, w) uint8 bitmaps by COCO's rasterisation rule (structures.py:548-556: frPyObjects + merge + decode)."""
        from ..evaluation import mask as mask_util
        if len(self.masks) == 0:
            return np.empty((0, self.height, self.width), dtype=np.uint8)
        out = []
        for obj in self.masks:
            rles = mask_util.frPyObjects([np.asarray(p, dtype=np.float64).tolist() for p in obj], self.height, self.width)
            out.append(mask_util.decode(mask_util.merge(rles)))
        return np.stack(out).reshape(-1, self.height, self.width)

    @property
    def areas(self):
        out = []
        for obj in self.masks:
            a = 0
            for p in obj:
                x, y = p[0::2], p[1::2]
                a += 0.5 * np.abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))
            out.append(a)
        return np.asarray(out)
