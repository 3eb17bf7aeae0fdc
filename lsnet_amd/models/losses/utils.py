"""Loss reduction helpers (mmdet/models/losses/utils.py:6-98)."""
import functools

import torch.nn.functional as F


def reduce_loss(loss, reduction):
    kind = F._Reduction.get_enum(reduction)  # 0 none, 1 mean, 2 sum
    return loss if kind == 0 else (loss.mean() if kind == 1 else loss.sum())


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """Element-wise weight, then reduce; with `avg_factor` the 'mean' is sum / avg_factor."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def weighted_loss(loss_func):
    """Turns `f(pred, target, **kw) -> elementwise loss` into
    `f(pred, target, weight=None, reduction='mean', avg_factor=None, **kw)`."""

    @functools.wraps(loss_func)
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        return weight_reduce_loss(loss_func(pred, target, **kwargs), weight, reduction, avg_factor)

    return wrapper
