"""GroupNorm with an optional fused ReLU, channels-last, several tensors per launch.

Stands where the reference uses torch.nn.GroupNorm (+ nn.ReLU): NORM_LAYERS['GN'] of ConvModule
(mmcv/cnn/bricks/norm.py:70-118), `DCNConvModule.bn` and `LSHead.{cls,bbox,...}_GN`
(lsnet_head.py:1830-1849, 136-141).  `GroupNorm` subclasses nn.GroupNorm (same parameters, same state-dict
keys, same definition: biased variance over HW x C/G, eps inside the root).  CUDA channels-last fp32 inputs of a
supported shape run the HIP kernels of csrc/norm.hip (lsn_group_norm_*), which keep the NHWC layout the deformable
convolutions want; anything else takes ATen's group_norm -- a different PyTorch operator for a different layout, not
a CPU fallback of the HIP path."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import grad_sink
from .backend import get_backend

_CL = torch.channels_last


class _GroupNormCatFn(torch.autograd.Function):
    """GroupNorm (+ ReLU) of the levels of a shared module whose outputs are wanted as LSHead._cat_px would lay them out: one
    (B, C, N_all, 1) tensor, the pixel rows of all levels back to back.  The kernels write the levels where they belong and
    read the levels of the gradient where they lie (lsn_gn_level.y_batch_stride / dy_batch_stride): no torch.cat behind the
    norm, no per-level copies of the gradient in front of its backward (12 launches per step in LSHead's two pixel-wise
    output branches, two of them over the 100 x 168 maps)."""

    @staticmethod
    def forward(ctx, gamma, beta, groups, eps, relu, *xs):
        be = get_backend(xs[0])
        y, mean_rstd = be.group_norm_forward(list(xs), gamma, beta, groups, eps, relu, cat_px=True)
        ctx.save_for_backward(gamma, beta, mean_rstd, *xs)
        ctx.cfg = (groups, relu)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        gamma, beta, mean_rstd, *xs = ctx.saved_tensors
        groups, relu = ctx.cfg
        need_p = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        be = get_backend(xs[0])
        sg = grad_sink.sink(gamma) if ctx.needs_input_grad[0] else None
        sb = grad_sink.sink(beta) if ctx.needs_input_grad[1] else None
        sinks = (sg, sb) if (sg is not None and sb is not None and getattr(be, 'supports_grad_sinks', False)) else None
        if sinks:
            dxs, _, _ = be.group_norm_backward(xs, dy, gamma, beta, groups, relu, mean_rstd, True, sinks=sinks)
            grad_sink.done(gamma)
            grad_sink.done(beta)
            return (None, None, None, None, None, *dxs)
        dxs, dg, db = be.group_norm_backward(xs, dy, gamma, beta, groups, relu, mean_rstd, need_p)
        return (dg if ctx.needs_input_grad[0] else None, db if ctx.needs_input_grad[1] else None, None, None, None, *dxs)


class _GroupNormFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, gamma, beta, groups, eps, relu, *xs):
        be = get_backend(xs[0])
        ys, mean_rstd = be.group_norm_forward(list(xs), gamma, beta, groups, eps, relu)
        ctx.save_for_backward(gamma, beta, mean_rstd, *xs)
        ctx.cfg = (groups, relu)
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *dys):
        gamma, beta, mean_rstd, *xs = ctx.saved_tensors
        groups, relu = ctx.cfg
        dys = [torch.zeros_like(x) if d is None else d for d, x in zip(dys, xs)]
        need_p = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        be = get_backend(xs[0])
        sg = grad_sink.sink(gamma) if ctx.needs_input_grad[0] else None
        sb = grad_sink.sink(beta) if ctx.needs_input_grad[1] else None
        sinks = (sg, sb) if (sg is not None and sb is not None and getattr(be, 'supports_grad_sinks', False)) else None
        if sinks:
            dxs, _, _ = be.group_norm_backward(xs, dys, gamma, beta, groups, relu, mean_rstd, True, sinks=sinks)
            grad_sink.done(gamma)
            grad_sink.done(beta)
            return (None, None, None, None, None, *dxs)
        dxs, dg, db = be.group_norm_backward(xs, dys, gamma, beta, groups, relu, mean_rstd, need_p)
        return (dg if ctx.needs_input_grad[0] else None, db if ctx.needs_input_grad[1] else None, None, None, None,
                *dxs)


def _hip_ok(x, C, G):
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=_CL)):
        return False
    return get_backend(x).group_norm_supported(C, G)


class GroupNorm(nn.GroupNorm):

    def forward_multi(self, xs, relu=False):
        """[GN(x) (then ReLU)] for tensors that share this module -- one fused launch on the HIP path."""
        xs = list(xs)
        if self.affine and all(_hip_ok(x, self.num_channels, self.num_groups) for x in xs):
            return list(_GroupNormFn.apply(self.weight, self.bias, self.num_groups, self.eps, bool(relu), *xs))
        if xs:
            from .conv import _warn_aten_fallback
            _warn_aten_fallback('GroupNorm', xs[0], f'{self.num_channels} channels, {self.num_groups} groups, affine {self.affine}, '
                                                    f'input {tuple(xs[0].shape)} {xs[0].dtype}, channels_last '
                                                    f'{xs[0].dim() == 4 and xs[0].is_contiguous(memory_format=_CL)}')
        ys = [F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps) for x in xs]
        return [F.relu(y) for y in ys] if relu else ys

    def forward_cat_px(self, xs, relu=False):
        """GN(x) (then ReLU) of the levels, returned as ONE (B, C, N_all, 1) tensor with the pixel rows of all levels back to
        back -- what `LSHead._cat_px(self.forward_multi(xs, relu))` returns, without the concatenation."""
        xs = list(xs)
        if self.affine and len(xs) <= 16 and all(_hip_ok(x, self.num_channels, self.num_groups) for x in xs) \
                and len({x.shape[0] for x in xs}) == 1 and getattr(get_backend(xs[0]), 'supports_grad_sinks', False):
            return _GroupNormCatFn.apply(self.weight, self.bias, self.num_groups, self.eps, bool(relu), *xs)
        ys = self.forward_multi(xs, relu)
        B, C = ys[0].shape[:2]
        x = torch.cat([m.permute(0, 2, 3, 1).reshape(B, -1, C) for m in ys], dim=1)
        return x.unsqueeze(2).permute(0, 3, 1, 2)

    def forward(self, x):
        return self.forward_multi([x], relu=False)[0]

    def forward_act(self, x):
        return self.forward_multi([x], relu=True)[0]
