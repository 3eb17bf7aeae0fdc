"""Frozen-statistics BatchNorm fused with the residual add and the ReLU that follow it in a ResNet block.

The reference computes `relu(bn(conv(x)) + identity)` with separate kernels (mmdet/models/backbones/resnet.py:261-301)
and keeps its BatchNorm layers in eval mode while training (`norm_eval=True`, :636-645).  `bn_act` gives the same
value through one HIP kernel each way (csrc/norm.hip, lsn_bn_eval_act_*) when the layer is in eval mode and the
tensor is CUDA / channels-last / fp32 / a supported channel count; otherwise it is the reference's sequence of
ATen ops (training-mode batch statistics are not on the LSNet path)."""
import ctypes

import torch
import torch.nn.functional as F

from .. import _lib
from . import grad_sink

_CL = torch.channels_last


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _BnActFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, mean, var, eps, relu):
        B, C, H, W = x.shape
        y = torch.empty_like(x, memory_format=_CL)
        _lib.check(_lib.load().lsn_bn_eval_act_forward(_p(x), _p(residual), _p(y), _p(mean), _p(var), _p(gamma),
                                                       _p(beta), ctypes.c_float(eps), 1 if relu else 0, B * H * W, C,
                                                       _stream()))
        ctx.save_for_backward(x, y if relu else None, gamma, mean, var)
        ctx.cfg = (eps, relu, residual is not None)
        ctx.beta_ref = beta    # backward only asks where its gradient goes (ops/grad_sink.py)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y, gamma, mean, var = ctx.saved_tensors
        eps, relu, has_res = ctx.cfg
        B, C, H, W = x.shape
        dy = dy.contiguous(memory_format=_CL)
        need_x, need_res = ctx.needs_input_grad[0], has_res and ctx.needs_input_grad[1]
        need_p = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dx = torch.empty_like(x, memory_format=_CL) if need_x else None
        dres = torch.empty_like(x, memory_format=_CL) if need_res else None
        sg = grad_sink.sink(gamma) if ctx.needs_input_grad[2] else None
        sb = grad_sink.sink(ctx.beta_ref) if ctx.needs_input_grad[3] else None
        acc = 1 if (sg is not None and sb is not None) else 0
        if acc:
            dg, db = sg, sb
        else:
            dg = torch.empty_like(gamma) if need_p else None
            db = torch.empty_like(gamma) if need_p else None
        lib = _lib.load()
        ws = torch.empty(lib.lsn_bn_eval_act_workspace_bytes(B * H * W, C), device=x.device, dtype=torch.uint8) \
            if need_p else None
        _lib.check(lib.lsn_bn_eval_act_backward(_p(dy), _p(y), _p(x), _p(mean), _p(var), _p(gamma), ctypes.c_float(eps),
                                                1 if relu else 0, _p(dx), _p(dres), _p(dg), _p(db), _p(ws), B * H * W, C,
                                                acc, _stream()))
        if acc:
            grad_sink.done(gamma)
            grad_sink.done(ctx.beta_ref)
            dg = db = None
        return dx, dres, dg if ctx.needs_input_grad[2] else None, db if ctx.needs_input_grad[3] else None, \
            None, None, None, None


def _hip_ok(bn, x, residual):
    if bn.training or not bn.affine or not bn.track_running_stats or not isinstance(bn, torch.nn.BatchNorm2d):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=_CL)):
        return False
    if residual is not None and not (residual.shape == x.shape and residual.dtype == torch.float32
                                     and residual.is_contiguous(memory_format=_CL)):
        return False
    C = x.shape[1]
    return C % 4 == 0 and (256 % (C // 4) == 0 if C <= 1024 else C % 1024 == 0)


def bn_act(bn, x, relu=False, residual=None):
    """act(bn(x) + residual) for a BatchNorm2d module `bn`."""
    if _hip_ok(bn, x, residual):
        return _BnActFn.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bool(relu))
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out
