"""ctypes front-end of oracle/lsnet_oracle.c (CPU restatement of the reference's native ops).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg and oracle/ref_harness.  Takes and returns CPU torch tensors (contiguous NCHW float32).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liblsnet_oracle.so')
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    src = os.path.join(_HERE, 'lsnet_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liblsnet_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_deform_conv_forward.restype = ctypes.c_int
        _lib.orc_deform_conv_backward.restype = ctypes.c_int
        _lib.orc_nms.restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(t):
    if t is None:
        return None
    assert t.device.type == 'cpu' and t.is_contiguous(), 'oracle takes contiguous CPU tensors'
    if t.dtype == torch.float32:
        return ctypes.cast(t.data_ptr(), _f)
    if t.dtype == torch.int64:
        return ctypes.cast(t.data_ptr(), _i64)
    raise TypeError(t.dtype)


def _c(t):
    return None if t is None else t.detach().float().contiguous()


def out_size(in_size, k, stride, pad, dil):
    return (in_size + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def deform_conv_forward(x, weight, bias, offset, mask, stride=1, pad=0, dil=1, groups=1, dg=1,
                        scale_h=1.0, scale_w=1.0, out_hw=None):
    """DCNv1 (mask None), DCNv2 (mask given) and pyramid DCN (out grid = offset grid, scales)."""
    x, weight, bias, offset, mask = _c(x), _c(weight), _c(bias), _c(offset), _c(mask)
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    if out_hw is None:
        out_hw = (offset.shape[2], offset.shape[3])
    Ho, Wo = out_hw
    assert offset.shape == (B, dg * 2 * kh * kw, Ho, Wo), (offset.shape, (B, dg * 2 * kh * kw, Ho, Wo))
    out = torch.empty(B, Co, Ho, Wo)
    rc = lib().orc_deform_conv_forward(
        _p(x), _p(weight), _p(bias), _p(offset), _p(mask), _p(out), B, C, H, W, Co, Ho, Wo, kh, kw,
        int(stride), int(pad), int(dil), int(groups), int(dg), ctypes.c_float(scale_h),
        ctypes.c_float(scale_w))
    assert rc == 0, rc
    return out


def deform_conv_backward(x, weight, offset, mask, gout, stride=1, pad=0, dil=1, groups=1, dg=1,
                         scale_h=1.0, scale_w=1.0, want=('gx', 'goff', 'gmask', 'gw', 'gb')):
    x, weight, offset, mask, gout = _c(x), _c(weight), _c(offset), _c(mask), _c(gout)
    B, C, H, W = x.shape
    Co, Cg, kh, kw = weight.shape
    Ho, Wo = gout.shape[2], gout.shape[3]
    res = {
        'gx': torch.zeros_like(x) if 'gx' in want else None,
        'goff': torch.zeros_like(offset) if 'goff' in want else None,
        'gmask': torch.zeros_like(mask) if ('gmask' in want and mask is not None) else None,
        'gw': torch.zeros_like(weight) if 'gw' in want else None,
        'gb': torch.zeros(Co) if 'gb' in want else None,
    }
    rc = lib().orc_deform_conv_backward(
        _p(x), _p(weight), _p(offset), _p(mask), _p(gout), _p(res['gx']), _p(res['goff']),
        _p(res['gmask']), _p(res['gw']), _p(res['gb']), B, C, H, W, Co, Ho, Wo, kh, kw, int(stride),
        int(pad), int(dil), int(groups), int(dg), ctypes.c_float(scale_h), ctypes.c_float(scale_w))
    assert rc == 0, rc
    return res


def sigmoid_focal_loss_forward(logits, targets, gamma, alpha):
    logits = _c(logits)
    targets = targets.detach().long().contiguous()
    N, C = logits.shape
    out = torch.empty_like(logits)
    lib().orc_sigmoid_focal_loss_forward(_p(logits), _p(targets), _p(out), N, C, ctypes.c_float(gamma),
                                         ctypes.c_float(alpha))
    return out


def sigmoid_focal_loss_backward(logits, targets, d_losses, gamma, alpha):
    logits, d_losses = _c(logits), _c(d_losses)
    targets = targets.detach().long().contiguous()
    N, C = logits.shape
    out = torch.empty_like(logits)
    lib().orc_sigmoid_focal_loss_backward(_p(logits), _p(targets), _p(d_losses), _p(out), N, C,
                                          ctypes.c_float(gamma), ctypes.c_float(alpha))
    return out


def nms(dets, thr):
    """dets (n,5) -> keep indices (int64) in descending score order (nms_cpu.cpp:8-66)."""
    dets = _c(dets)
    n = dets.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64)
    k = lib().orc_nms(_p(dets), n, ctypes.c_float(thr), _p(keep))
    return keep[:k].clone()


def num_threads():
    return lib().orc_num_threads()


# ---------------------------------------------------------------------------------------------
# autograd wrappers so the oracle can stand in for the native ops on CPU tensors in tests and in
# bench.py's cpu_baseline leg (registered as the 'cpu' backend of lsnet_amd.ops by those callers).
# ---------------------------------------------------------------------------------------------
class _DeformConvFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, pad, dil, groups, dg, scale_h, scale_w):
        ctx.cfg = (stride, pad, dil, groups, dg, scale_h, scale_w)
        ctx.has_mask, ctx.has_bias = mask is not None, bias is not None
        ctx.save_for_backward(x, offset, mask if mask is not None else x.new_empty(0), weight)
        return deform_conv_forward(x, weight, bias, offset, mask, stride, pad, dil, groups, dg, scale_h,
                                   scale_w)

    @staticmethod
    def backward(ctx, gout):
        x, offset, mask, weight = ctx.saved_tensors
        stride, pad, dil, groups, dg, sh, sw = ctx.cfg
        g = deform_conv_backward(x, weight, offset, mask if ctx.has_mask else None, gout.contiguous(),
                                 stride, pad, dil, groups, dg, sh, sw)
        return (g['gx'], g['goff'], g['gmask'] if ctx.has_mask else None, g['gw'],
                g['gb'] if ctx.has_bias else None, None, None, None, None, None, None, None)


def deform_conv(x, offset, mask, weight, bias, stride=1, pad=0, dil=1, groups=1, dg=1, scale_h=1.0,
                scale_w=1.0):
    return _DeformConvFn.apply(x, offset, mask, weight, bias, stride, pad, dil, groups, dg,
                               float(scale_h), float(scale_w))


class _FocalFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        ctx.ga = (gamma, alpha)
        return sigmoid_focal_loss_forward(logits, targets, gamma, alpha)

    @staticmethod
    def backward(ctx, d):
        logits, targets = ctx.saved_tensors
        return sigmoid_focal_loss_backward(logits, targets, d.contiguous(), *ctx.ga), None, None, None


def sigmoid_focal_loss(logits, targets, gamma=2.0, alpha=0.25):
    return _FocalFn.apply(logits, targets, gamma, alpha)
