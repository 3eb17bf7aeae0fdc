"""Sigmoid focal loss op -- mirror of mmdet/ops/sigmoid_focal_loss/sigmoid_focal_loss.py:8-54."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .backend import get_backend


class SigmoidFocalLossFunction(Function):

    @staticmethod
    def forward(ctx, input, target, gamma=2.0, alpha=0.25):
        ctx.save_for_backward(input, target)
        ctx.gamma, ctx.alpha = float(gamma), float(alpha)
        return get_backend(input).focal_forward(input, target, ctx.gamma, ctx.alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        input, target = ctx.saved_tensors
        d_input = get_backend(input).focal_backward(input, target, d_loss.contiguous(), ctx.gamma, ctx.alpha)
        return d_input, None, None, None


sigmoid_focal_loss = SigmoidFocalLossFunction.apply


class _FocalSumFunction(Function):
    """sum_n weight[n] * sum_c FL(n, c) as ONE reduction kernel; the backward is one elementwise
    kernel that reads the upstream scalar gradient from device memory (no host sync)."""

    @staticmethod
    def forward(ctx, input, target, weight, gamma, alpha):
        ctx.save_for_backward(input, target, weight if weight is not None else input.new_empty(0))
        ctx.has_w, ctx.gamma, ctx.alpha = weight is not None, float(gamma), float(alpha)
        return get_backend(input).focal_sum(input, target, weight, ctx.gamma, ctx.alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_sum):
        input, target, weight = ctx.saved_tensors
        d_input = get_backend(input).focal_backward_weighted(input, target, weight if ctx.has_w else None,
                                                             d_sum.reshape(1), ctx.gamma, ctx.alpha)
        return d_input, None, None, None, None


def sigmoid_focal_loss_sum(input, target, weight=None, gamma=2.0, alpha=0.25):
    return _FocalSumFunction.apply(input, target, weight, gamma, alpha)


class _FocalLevelSumsFunction(Function):
    """(L,) per-level sums of weight[n] * sum_c FL(n, c) over LSHead's concatenated rows (B images x N_all rows, the levels back to
    back in every image): one launch for all levels forward and one backward (lsn_sigmoid_focal_loss_level_sums)."""

    @staticmethod
    def forward(ctx, input, target, weight, B, num_level, gamma, alpha):
        ctx.save_for_backward(input, target, weight if weight is not None else input.new_empty(0))
        ctx.has_w, ctx.cfg = weight is not None, (int(B), tuple(int(n) for n in num_level), float(gamma), float(alpha))
        return get_backend(input).focal_level_sums(input, target, weight, *ctx.cfg)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_sums):
        input, target, weight = ctx.saved_tensors
        B, num_level, gamma, alpha = ctx.cfg
        d_input = get_backend(input).focal_backward_levels(input, target, weight if ctx.has_w else None, d_sums, B, num_level,
                                                           gamma, alpha)
        return d_input, None, None, None, None, None, None


def level_rows_ok(t, B, num_level):
    """Can the per-level kernels take this tensor?  (device fp32, at most 8 levels and 64 (image, level) ranges)"""
    return t.is_cuda and t.dtype == torch.float32 and len(num_level) <= 8 and B * len(num_level) <= 64


def sigmoid_focal_loss_level_sums(input, target, weight, B, num_level, gamma=2.0, alpha=0.25):
    return _FocalLevelSumsFunction.apply(input, target, weight, B, num_level, gamma, alpha)


class _LevelSumsFunction(Function):
    """(L,) per-level sums of per-row values, the same layout (lsn_level_sums / lsn_level_expand)."""

    @staticmethod
    def forward(ctx, rows, B, num_level):
        ctx.cfg = (int(B), tuple(int(n) for n in num_level))
        return get_backend(rows).level_sums(rows.contiguous(), *ctx.cfg)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return get_backend(g).level_expand(g, *ctx.cfg), None, None


def level_sums(rows, B, num_level):
    """rows: (B * N_all,) -> (L,)"""
    return _LevelSumsFunction.apply(rows, B, num_level)


class SigmoidFocalLoss(nn.Module):

    def __init__(self, gamma, alpha):
        super().__init__()
        self.gamma, self.alpha = gamma, alpha

    def forward(self, logits, targets):
        assert logits.is_cuda
        return sigmoid_focal_loss(logits, targets, self.gamma, self.alpha).sum()

    def __repr__(self):
        return f'{self.__class__.__name__}(gamma={self.gamma}, alpha={self.alpha})'
