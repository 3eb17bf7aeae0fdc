"""Small helpers of the target pipeline (mmdet/core/utils/misc.py:35-67, core/anchor/utils.py:4-17)."""
from functools import partial

import torch


def multi_apply(func, *args, **kwargs):
    """Map `func` over zipped argument lists and transpose the per-call result tuples into a
    tuple of lists."""
    fn = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(fn, *args))))


def unmap(data, count, inds, fill=0):
    """Scatter `data` (one row per True in `inds`) back into `count` rows filled with `fill`."""
    shape = (count,) + tuple(data.shape[1:])
    out = data.new_full(shape, fill)
    out[inds.bool()] = data
    return out


def images_to_levels(target, num_levels):
    """[per-image (sum(N_l), ...)] -> [per-level (num_imgs, N_l, ...)]"""
    stacked = torch.stack(target, 0)
    out, start = [], 0
    for n in num_levels:
        out.append(stacked[:, start:start + n])
        start += n
    return out
