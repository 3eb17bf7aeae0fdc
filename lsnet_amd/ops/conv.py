"""Dense convolution behind CONV_LAYERS['Conv2d'] (the reference: torch.nn.Conv2d through
mmcv/cnn/bricks/conv.py:11-43 `build_conv_layer`).

`Conv2d` subclasses nn.Conv2d (same parameters and state-dict keys).  CUDA channels-last fp32 inputs of a
supported shape run the split-bf16 implicit-GEMM kernels of csrc/conv.hip when the library's math mode is
'bf16x6' (fp32-equivalent, the default) or 'bf16x3'; everything else -- exact-fp32 mode, grouped / odd shapes, CPU tensors -- goes to
ATen's convolution (MIOpen), which is a different vendor operator, not a fallback of the HIP path."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

_CL = torch.channels_last


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ConvFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, relu):
        lib = _lib.load()
        B, C, H, W = x.shape
        Co, _, kh, kw = w.shape
        Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
        w = w.contiguous(memory_format=_CL)
        out = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
        ws = torch.empty(2 * w.numel(), device=x.device, dtype=torch.float32)
        _lib.check(lib.lsn_conv2d_forward(_p(x), _p(w), _p(bias), _p(out), _p(ws), B, H, W, C, Co, kh, kw, stride, pad,
                                          dil, 1 if relu else 0, _stream()))
        ctx.save_for_backward(x, w, out if relu else None)
        ctx.cfg = (stride, pad, dil, relu, bias is not None)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        x, w, out = ctx.saved_tensors
        stride, pad, dil, relu, has_bias = ctx.cfg
        lib = _lib.load()
        go = go.contiguous(memory_format=_CL)
        if relu:
            go = go * (out > 0)
        B, C, H, W = x.shape
        Co, _, kh, kw = w.shape
        gx = gw = gb = None
        own_data = (stride == 1 and Co % 4 == 0 and dil * (kh - 1) - pad == dil * (kw - 1) - pad >= 0
                    and _own_is_faster(B * H * W, Co, C, kh * kw))
        if ctx.needs_input_grad[0] and own_data:
            gx = torch.empty_like(x, memory_format=_CL)
            ws = torch.empty(2 * w.numel(), device=x.device, dtype=torch.float32)
            _lib.check(lib.lsn_conv2d_backward_data(_p(go), _p(w), _p(gx), _p(ws), B, H, W, C, Co, kh, kw, stride, pad,
                                                    dil, _stream()))
        own_w = ctx.needs_input_grad[1] and Co >= 256 and C >= 256 and C * kh * kw >= 512 and C % 4 == 0
        if own_w:   # weight gradient (and the bias gradient in the same pass) through the split-bf16 kernel
            gw = torch.empty_like(w)
            want_b = has_bias and ctx.needs_input_grad[2]
            gb = torch.empty(Co, device=x.device, dtype=torch.float32) if want_b else None
            _lib.check(lib.lsn_conv2d_backward_weight(_p(x), _p(go), _p(gw), _p(gb), B, H, W, C, Co, kh, kw, stride, pad,
                                                      dil, _stream()))
        need_aten = (ctx.needs_input_grad[0] and gx is None, ctx.needs_input_grad[1] and not own_w, False)
        if need_aten[0] or need_aten[1]:
            ax, aw, _ = torch.ops.aten.convolution_backward(go, x, w, None, [stride, stride], [pad, pad], [dil, dil],
                                                            False, [0, 0], 1, list(need_aten))
            gx = ax if need_aten[0] else gx
            gw = aw if need_aten[1] else gw
        if has_bias and ctx.needs_input_grad[2] and gb is None:
            gb = go.sum(dim=(0, 2, 3))
        return gx, gw, gb, None, None, None, None


def hip_conv_ok(x, weight, stride, padding, dilation, groups, padding_mode='zeros'):
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=_CL)):
        return False
    if groups != 1 or padding_mode != 'zeros' or isinstance(padding, str):
        return False
    if stride[0] != stride[1] or padding[0] != padding[1] or dilation[0] != dilation[1]:
        return False
    C = x.shape[1]
    if C % 4 != 0 or C < 16 or x.numel() * 4 >= 2 ** 31 or weight.numel() * 4 >= 2 ** 31:
        return False
    return _lib.split_math()


def _own_is_faster(pixels, cin, cout, taps):
    """Shape rule from tools/bench_convs_x3.py (MI355X, ROCm 7.2 MIOpen): the split-bf16 kernel wins once its grid
    covers a good part of the chip (>= 100 blocks of 64x256 / 128x128 / 256x64) and the reduction is deep enough to amortise the
    pipeline prologue; MIOpen keeps the small and shallow layers."""
    bm, bn = (256, 64) if cout <= 64 else ((128, 128) if cout <= 128 else (64, 256))
    blocks = -(-pixels // bm) * -(-cout // bn)
    return blocks >= 100 and cin * taps >= 256 and cin >= 128 and cout >= 128


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False):
    return _ConvFn.apply(x, weight, bias, int(stride), int(padding), int(dilation), bool(relu))


class Conv2d(nn.Conv2d):

    def forward(self, x, weight=None):
        """`weight`: use this tensor instead of the module's parameter (a zero-padded view of it, for inputs whose
        channel count was rounded up to the kernels' granularity)."""
        if weight is not None:
            return self._run(x, weight)
        return self._run(x, self.weight)

    def _run(self, x, w):
        if hip_conv_ok(x, w, self.stride, self.padding, self.dilation, self.groups, self.padding_mode):
            Co, C, kh, kw = w.shape
            s, p, d = self.stride[0], self.padding[0], self.dilation[0]
            ho = (x.shape[2] + 2 * p - (d * (kh - 1) + 1)) // s + 1
            wo = (x.shape[3] + 2 * p - (d * (kw - 1) + 1)) // s + 1
            if _own_is_faster(x.shape[0] * ho * wo, C, Co, kh * kw):
                return _ConvFn.apply(x, w, self.bias, s, p, d, False)
        return F.conv2d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)
