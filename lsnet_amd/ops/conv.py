"""Dense convolution behind CONV_LAYERS['Conv2d'] (the reference: torch.nn.Conv2d through
mmcv/cnn/bricks/conv.py:11-43 `build_conv_layer`).

`Conv2d` subclasses nn.Conv2d (same parameters and state-dict keys).  Every dense (groups = 1) convolution of a CUDA
channels-last fp32 tensor -- forward, data gradient (any stride), weight and bias gradient -- runs the split-bf16
implicit-GEMM kernels of csrc/conv_kernels.h in the library's math modes 'bf16x6' (fp32-equivalent, the default) and
'bf16x3'.  The kernels read weights as prepared images (MFMA fragment order); `weight_image` keeps one per (weight
tensor, pass) and rebuilds it when the tensor's version counter moves, i.e. once per optimizer step.  Grouped
convolutions with 4 .. 32 channels per group (ResNeXt 64x4d bottlenecks) run the exact-fp32 kernels of csrc/gconv.hip
in every math mode.  Contiguous (NCHW) inputs are re-laid channels-last and LSN_MATH_FP32 runs the same fp32-equivalent
kernels: no vendor convolution is reached from a CUDA fp32 tensor.  Other group shapes, dtypes and CPU tensors go to ATen's
convolution, which is a different operator, not a fallback of the HIP path."""
import ctypes
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from . import grad_sink
from .streams import side_stream

_CL = torch.channels_last


_warned_fallbacks = set()


def _warn_aten_fallback(what, x, detail):
    """A device tensor that leaves the library's own kernels for ATen (MIOpen on the device): legal -- shapes the kernels do
    not serve, tensors of 2 GiB and more, other dtypes -- but never silent.  Once per (operator, reason)."""
    if not (torch.is_tensor(x) and x.is_cuda):
        return
    key = (what, detail)
    if key in _warned_fallbacks:
        return
    _warned_fallbacks.add(key)
    import warnings
    warnings.warn(f'lsnet_amd.{what}: no HIP kernel of this library serves this call on {x.device} ({detail}); '
                  'falling back to the ATen operator', RuntimeWarning, stacklevel=3)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pad_channels(t, c_to):
    """Zero-pad dim 1 of a 4-D tensor to `c_to` channels, keeping channels-last memory."""
    out = t.new_zeros((t.shape[0], c_to, t.shape[2], t.shape[3])).contiguous(memory_format=_CL)
    out[:, :t.shape[1]] = t
    return out


_images = {}    # (id(weight), kind, stride, pad, dil, math mode, id(bn) or 0) -> [weakref, version, data_ptr, image,
#                  optimizer epoch, weakref(bn) or None, shift tensor or None, bn versions]
_opt_epoch = [0]


# LSNET_SIDE_STREAM_IMAGES=0: the images are rebuilt on the stream that asks for them (A/B switch)
SIDE_STREAM_IMAGES = os.environ.get('LSNET_SIDE_STREAM_IMAGES', '1') != '0'
_step_done = {}      # device index -> event recorded right behind the last optimizer step (on the stream that ran it)


def _after_optimizer_step(*_):
    """Global optimizer post-step hook: every image of a trainable weight is stale now.  (The tensors' version counters
    are not enough: torch's FUSED optimizers update parameters without bumping them.)"""
    _opt_epoch[0] += 1
    if SIDE_STREAM_IMAGES and torch.cuda.is_available() and torch.cuda.is_initialized() \
            and not torch.cuda.is_current_stream_capturing():
        dev = torch.cuda.current_device()
        ev = _step_done.get(dev)
        if ev is None:
            ev = _step_done[dev] = torch.cuda.Event()
        ev.record()


def parameters_updated():
    """For optimizers that do not derive from torch.optim.Optimizer (runner/fused_sgd.py writes the parameters from its own
    kernel): the trainable weights moved -- what the global post-step hook tells the cache for torch's optimizers."""
    _after_optimizer_step()


from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_step  # noqa: E402

_reg_post_step(_after_optimizer_step)


def invalidate_weight_images():
    """Mark EVERY cached image stale (frozen weights included).  Staleness is otherwise inferred from the tensors' version
    counters, their storage pointers and the global optimizer post-step hook -- in-place updates through `p.data`
    (checkpoint loading, the parameter broadcast of DataParallelModel, EMA / weight-surgery hooks, optimizers that do not
    derive from torch.optim.Optimizer) are invisible to all three: whoever writes parameters that way calls this.
    GraphedForwardBackward also calls it right before a capture, so that the image-rebuild launches are ALWAYS recorded
    into the graph (captured with fresh images, every replay would run forward and data gradient on the capture-time
    weights while the optimizer keeps moving them)."""
    _opt_epoch[0] += 1
    _step_done.clear()      # (the writes this call stands for came after the optimizer step: no rebuild behind its event)
    for ent in _images.values():
        ent[1] = -1


def _bn_versions(bn):
    return (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
            bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps))


def _stale(ent, w):
    if ent[1] != w._version or ent[2] != w.data_ptr():
        return True
    bn = ent[5]() if ent[5] is not None else None
    if ent[5] is not None and (bn is None or ent[7] != _bn_versions(bn)):
        return True
    trainable = w.requires_grad or (bn is not None and (bn.weight.requires_grad or bn.bias.requires_grad))
    return trainable and ent[4] != _opt_epoch[0]


def _fill_item(it, key, ent, w):
    Co, C, kh, kw = w.shape
    it.kind, it.w, it.prepared = key[1], w.data_ptr(), ent[3].data_ptr()
    it.C, it.Co, it.kh, it.kw, it.stride, it.pad, it.dil = C, Co, kh, kw, key[2], key[3], key[4]
    bn = ent[5]() if ent[5] is not None else None
    if bn is not None:   # a BatchNorm folded into the image (lsn_conv_wprep); the shift belongs to the forward image
        it.bn_gamma, it.bn_var, it.bn_beta, it.bn_mean = bn.weight.data_ptr(), bn.running_var.data_ptr(), \
            bn.bias.data_ptr(), bn.running_mean.data_ptr()
        it.shift_out, it.bn_eps = (ent[6].data_ptr() if ent[6] is not None else None), float(bn.eps)


def _mark_fresh(ent, w):
    ent[1], ent[2], ent[4] = w._version, w.data_ptr(), _opt_epoch[0]
    if ent[5] is not None and ent[5]() is not None:
        ent[7] = _bn_versions(ent[5]())


def _refresh_stale(mode):
    """Rebuild EVERY cached image whose weight (or folded BatchNorm) has moved on -- all trainable convolutions after an
    optimizer step -- in one launch (lsn_conv2d_prepare_weights_multi): called when the first stale image of a step is
    asked for."""
    items = []
    for key, ent in _images.items():
        w = ent[0]()
        if w is None or key[5] != mode or (ent[5] is not None and ent[5]() is None) or not _stale(ent, w):
            continue
        items.append((key, ent, w))
    if not items:
        return
    lib = _lib.load()
    if torch.cuda.is_current_stream_capturing():
        # the multi-tensor launch uploads its job table when the set of stale images changed (a synchronous copy: not
        # allowed while a hipGraph is being captured): one launch per image instead, recorded into the graph
        for key, ent, w in items:
            it = _lib.ConvWprep()
            _fill_item(it, key, ent, w)
            _lib.check(lib.lsn_conv2d_prepare_weights_item(ctypes.byref(it), _stream()))
    else:
        arr = (_lib.ConvWprep * len(items))()
        for it, (key, ent, w) in zip(arr, items):
            _fill_item(it, key, ent, w)
        dev = items[0][2].device.index
        ev = _step_done.get(dev) if SIDE_STREAM_IMAGES and len(items) > 1 else None
        if ev is not None and not all(w.device.index == dev and ent[1] == w._version and ent[2] == w.data_ptr() and
                                      (ent[5] is None or ent[7] == _bn_versions(ent[5]())) for _, ent, w in items):
            # Behind the event only what the event is known to cover: every image here must be stale for ONE reason, the optimizer
            # epoch -- same tensor version and storage as when it was built, so that no torch operation has written the weight since
            # (any in-place write bumps the version) and the writer was the library's own optimizer kernel (runner/fused_sgd.py calls
            # parameters_updated() right behind it).  torch's optimizers bump the versions: their steps rebuild in stream order.
            ev = None
        if ev is not None:
            # The rebuild (one bandwidth-bound launch, ~0.2 ms for LSNet R-50) needs the optimizer step and nothing later: it runs on
            # a second stream behind the event recorded there, beside whatever the asking stream has queued since -- the frozen stem
            # and first stage of the next forward, whose images never change -- and the asking stream waits for it here.
            main = torch.cuda.current_stream(dev)
            side = side_stream(torch.device('cuda', dev))       # (ONE per process and device: ops/streams.py says why)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                _lib.check(lib.lsn_conv2d_prepare_weights_multi(len(items), arr, _stream()))
            main.wait_stream(side)
        else:
            _lib.check(lib.lsn_conv2d_prepare_weights_multi(len(items), arr, _stream()))
    for key, ent, w in items:
        _mark_fresh(ent, w)


def weight_image(w, kind, stride=1, pad=0, dil=1, bn=None):
    """The prepared image (lsn_conv2d_prepare_weights) of a channels-last (Co, C, kh, kw) weight for the forward
    (kind 0) or backward-data (kind 1) pass.  Cached per tensor OBJECT; stale when the tensor's version counter or
    storage moved, or -- for a trainable weight -- when any optimizer has stepped since.  The stale images of all
    parameters are rebuilt together, once per optimizer step, when the first of them is needed; a temporary (padded
    view, test tensor) gets a fresh one and drops it when it dies.
    bn: an eval-mode BatchNorm2d folded into the image (every weight of forward output channel co scaled by gamma /
    sqrt(var + eps)).  kind 0 returns (image, shift) -- the shift goes into the convolution's bias slot; kind 1 returns
    the backward-data image of the scaled weight."""
    lib = _lib.load()
    mode = lib.lsn_get_math_mode()
    key = (id(w), kind, stride, pad, dil, mode, id(bn) if bn is not None else 0)
    ent = _images.get(key)
    if ent is not None and ent[0]() is w and (bn is None or (ent[5] is not None and ent[5]() is bn)):
        if _stale(ent, w):
            _refresh_stale(mode)
        return ent[3] if (bn is None or kind == 1) else (ent[3], ent[6])
    Co, C, kh, kw = w.shape
    nbytes = lib.lsn_conv2d_prepared_bytes(kind, C, Co, kh, kw, stride, pad, dil)
    if nbytes < 0:
        _lib.check(-2)
    img = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    shift = torch.empty(Co, device=w.device, dtype=torch.float32) if (bn is not None and kind == 0) else None
    ent = [weakref.ref(w, lambda _, k=key: _images.pop(k, None)), w._version, w.data_ptr(), img, _opt_epoch[0],
           weakref.ref(bn) if bn is not None else None, shift, _bn_versions(bn) if bn is not None else None]
    it = _lib.ConvWprep()
    _fill_item(it, key, ent, w)
    _lib.check(lib.lsn_conv2d_prepare_weights_item(ctypes.byref(it), _stream()))
    _images[key] = ent
    return img if (bn is None or kind == 1) else (img, shift)


def _levels(n):
    return (_lib.ConvLevel * n)()


def _param_grad_buffers(w, bias, want_w, want_b):
    """Where a weight / bias gradient kernel writes: the parameters' gradient sinks (ops/grad_sink.py; accumulate = 1)
    when BOTH wanted gradients have one, fresh tensors otherwise.  Returns (grad_w buffer, grad_b buffer or None, acc)."""
    sw = grad_sink.sink(w) if want_w else None
    sb = grad_sink.sink(bias) if (want_b and bias is not None) else None
    if want_w and sw is not None and (not want_b or sb is not None) and sw.is_contiguous(memory_format=_CL) == \
            w.is_contiguous(memory_format=_CL) and sw.stride() == w.stride():
        return sw, sb, 1
    gw = torch.empty_like(w)
    gb = torch.empty(w.shape[0], device=w.device, dtype=torch.float32) if want_b else None
    return gw, gb, 0


def _param_grad_results(w, bias, gw, gb, acc, want_w):
    """What the Function returns to autograd for (weight, bias) after the kernel ran."""
    if acc:
        grad_sink.done(w)
        if gb is not None:
            grad_sink.done(bias)
        return None, None
    return (gw if want_w else None), gb


# ---- kernel-level primitives (one C call each; the autograd Functions below and ops/resblock.py compose them) ----------
# LSNET_GATE_MULTI=0: one gate launch per map, on dense copies of sliced gradients (A/B switch)
GATE_MULTI = os.environ.get('LSNET_GATE_MULTI', '1') != '0'


def _images_apart(g):
    """Is g (B, C, H, W) a stack of dense channels-last images (whatever lies between them)?  -> floats between two images, or 0"""
    B, C, H, W = g.shape
    if g.dtype != torch.float32 or not g.is_cuda or g.data_ptr() % 16 or (C * H * W) % 4:
        return 0
    sb, sc, sh, sw = g.stride()
    if (C == 1 or sc == 1) and (W == 1 or sw == C) and (H == 1 or sh == W * C) and (B == 1 or (sb >= C * H * W and sb % 4 == 0)):
        return sb if B > 1 else C * H * W
    return 0


def relu_gate_multi(gys, ys):
    """[grad_y where y > 0 else 0] for the maps of one multi-level convolution in ONE launch (lsn_relu_gate_multi).  A gradient
    whose images lie apart -- a level sliced out of the head's concatenated tensor -- is read where it lies: no copy first."""
    n = len(gys)
    strides = [_images_apart(g) for g in gys]
    if not (0 < n <= 8 and all(strides) and all(y.is_contiguous(memory_format=_CL) and y.numel() % 4 == 0 and y.shape == g.shape
                                                 for g, y in zip(gys, ys))):
        gys = [g.contiguous(memory_format=_CL) for g in gys]
        return [relu_gate(g, o) if o.numel() % 4 == 0 else g * (o > 0) for g, o in zip(gys, ys)]
    outs = [torch.empty_like(y) for y in ys]
    jobs = (_lib.GateJob * n)()
    for j, (g, y, o, sb) in enumerate(zip(gys, ys, outs, strides)):
        jobs[j].grad_y, jobs[j].y, jobs[j].grad = g.data_ptr(), y.data_ptr(), o.data_ptr()
        jobs[j].B, jobs[j].per_image, jobs[j].gy_batch_stride = y.shape[0], y.numel() // y.shape[0], sb
    _lib.check(_lib.load().lsn_relu_gate_multi(n, jobs, _stream()))
    return outs


def relu_gate(gy, y):
    """grad_y where y > 0, else 0 -- the gradient through a ReLU whose output was stored (lsn_relu_gate)."""
    g = torch.empty_like(y, memory_format=_CL)
    _lib.check(_lib.load().lsn_relu_gate(_p(gy), _p(y), _p(g), ctypes.c_int64(y.numel()), _stream()))
    return g


def dgrad(g, w, in_shape, stride, pad, dil, bn=None, residual=None, gate=None, out=None):
    """Data gradient of conv(x, w [scaled by the folded eval-mode BatchNorm `bn`]) under the upstream gradient g
    (channels-last).  In the epilogue of the launch: `residual` (another path's gradient of the same tensor; may be
    `out` itself) is added and `gate` (the stored ReLU output whose gradient this is) zeroes the closed elements."""
    B, C, H, W = in_shape
    Co, _, kh, kw = w.shape
    if out is None:
        out = torch.empty(in_shape, device=g.device, dtype=torch.float32, memory_format=_CL)
    lv = _levels(1)
    lv[0].x, lv[0].out, lv[0].B, lv[0].H, lv[0].W = _p(g), _p(out), B, H, W
    lv[0].residual, lv[0].gate = _p(residual), _p(gate)
    _lib.check(_lib.load().lsn_conv2d_backward_data_prepared(1, lv, _p(weight_image(w, 1, stride, pad, dil, bn=bn)), C, Co, kh,
                                                             kw, stride, pad, dil, _stream()))
    return out


def wgrad_bn(x, g, w, bn, stride, pad, dil, need=(True, True, True)):
    """Parameter gradients of bn_eval(conv(x, w)) under g = the (gated) gradient w.r.t. the normalised output
    (lsn_conv2d_backward_weight_bn: exact for every gamma).  Accumulates into the three gradient sinks when all three
    parameters have one (returns (None, None, None) to autograd), else returns fresh (grad_w, grad_gamma, grad_beta)
    with None for the entries `need` does not ask for."""
    B, C, H, W = x.shape
    Co, _, kh, kw = w.shape
    gamma, beta = bn.weight, bn.bias
    sw, sg, sb = grad_sink.sink(w), grad_sink.sink(gamma), grad_sink.sink(beta)
    acc = 1 if (all(need) and sw is not None and sg is not None and sb is not None and sw.stride() == w.stride()) else 0
    if acc:
        gw, dg, db = sw, sg, sb
    else:
        gw, dg, db = torch.empty_like(w), torch.empty_like(gamma), torch.empty_like(beta)
    _lib.check(_lib.load().lsn_conv2d_backward_weight_bn(
        _p(x), _p(g), _p(w), _p(gamma), _p(bn.running_mean), _p(bn.running_var), ctypes.c_float(float(bn.eps)), _p(gw),
        _p(dg), _p(db), B, H, W, C, Co, kh, kw, stride, pad, dil, acc, _stream()))
    if acc:
        grad_sink.done(w)
        grad_sink.done(gamma)
        grad_sink.done(beta)
        return None, None, None
    return (gw if need[0] else None), (dg if need[1] else None), (db if need[2] else None)


def wgrad_bn_deferrable(w, bn):
    """Can this layer's parameter gradients be produced LATER, by a launch shared with other layers (wgrad_bn_jobs)?  Only
    into gradient sinks: autograd wants a returned gradient now."""
    sw = grad_sink.sink(w)
    return sw is not None and grad_sink.sink(bn.weight) is not None and grad_sink.sink(bn.bias) is not None \
        and sw.stride() == w.stride() and w.is_cuda


def wgrad_bn_jobs(jobs, stride, pad, dil):
    """jobs: [(x, g, w, bn)] of ONE geometry, every parameter with a sink (wgrad_bn_deferrable): one launch
    (lsn_conv2d_backward_weight_bn_jobs) adds every job's (grad_w, grad_gamma, grad_beta) to its sinks."""
    x0, _, w0, _ = jobs[0]
    B, C, H, W = x0.shape
    Co, _, kh, kw = w0.shape
    lib = _lib.load()
    for i in range(0, len(jobs), 8):
        part = jobs[i:i + 8]
        arr = (_lib.WgradBnJob * len(part))()
        for q, (x, g, w, bn) in zip(arr, part):
            assert x.shape == x0.shape and w.shape == w0.shape
            q.x, q.g, q.w = _p(x), _p(g), _p(w)
            q.bn_gamma, q.bn_mean, q.bn_var, q.bn_eps = _p(bn.weight), _p(bn.running_mean), _p(bn.running_var), float(bn.eps)
            q.grad_w, q.grad_gamma, q.grad_beta = _p(grad_sink.sink(w)), _p(grad_sink.sink(bn.weight)), _p(grad_sink.sink(bn.bias))
        _lib.check(lib.lsn_conv2d_backward_weight_bn_jobs(len(part), arr, B, H, W, C, Co, kh, kw, stride, pad, dil, 1, _stream()))
    for _, _, w, bn in jobs:
        grad_sink.done(w)
        grad_sink.done(bn.weight)
        grad_sink.done(bn.bias)


def conv_fwd_bn(x, w, bn, stride, pad, dil, relu, residual=None):
    """act(bn_eval(conv(x, w)) + residual) in one launch (the norm folded into the prepared image)."""
    B, C, H, W = x.shape
    Co, _, kh, kw = w.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    img, shift = weight_image(w, 0, bn=bn)
    out = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
    lv = _levels(1)
    lv[0].x, lv[0].out, lv[0].B, lv[0].H, lv[0].W, lv[0].residual = _p(x), _p(out), B, H, W, _p(residual)
    _lib.check(_lib.load().lsn_conv2d_forward_prepared(1, lv, _p(img), _p(shift), C, C, Co, kh, kw, stride, pad, dil,
                                                       1 if relu else 0, _stream()))
    return out


class _ConvFn(torch.autograd.Function):
    """Forward, data gradient (any stride: residue classes of the transposed convolution), weight and bias gradient:
    all through liblsnet_hip.so.  Needs C % 4 == 0 (`conv2d` pads the 3-channel stem input)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, relu):
        lib = _lib.load()
        B, C, H, W = x.shape
        Co, _, kh, kw = w.shape
        Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
        w = w.contiguous(memory_format=_CL)
        out = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
        lv = _levels(1)
        lv[0].x, lv[0].out, lv[0].B, lv[0].H, lv[0].W = _p(x), _p(out), B, H, W
        _lib.check(lib.lsn_conv2d_forward_prepared(1, lv, _p(weight_image(w, 0)), _p(bias), C, C, Co, kh, kw, stride, pad,
                                                   dil, 1 if relu else 0, _stream()))
        ctx.save_for_backward(x, w, out if relu else None)
        ctx.cfg = (stride, pad, dil, relu, bias is not None)
        ctx.bias_ref = bias    # (not saved for its value: backward only asks where its gradient goes)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        x, w, out = ctx.saved_tensors
        stride, pad, dil, relu, has_bias = ctx.cfg
        lib = _lib.load()
        go = go.contiguous(memory_format=_CL)
        if relu:
            go = relu_gate(go, out) if out.numel() % 4 == 0 else go * (out > 0)
        B, C, H, W = x.shape
        Co, _, kh, kw = w.shape
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            go8, w8, Co8 = go, w, Co
            if Co % 4 and C <= 64:    # (narrow tiles read grad_output in 4-channel pieces: zero filters are free)
                Co8 = (Co + 3) // 4 * 4
                go8 = _pad_channels(go, Co8)
                w8 = w.new_zeros((Co8, C, kh, kw)).contiguous(memory_format=_CL)
                w8[:Co] = w
            gx = torch.empty_like(x, memory_format=_CL)
            lv = _levels(1)
            lv[0].x, lv[0].out, lv[0].B, lv[0].H, lv[0].W = _p(go8), _p(gx), B, H, W
            _lib.check(lib.lsn_conv2d_backward_data_prepared(1, lv, _p(weight_image(w8, 1, stride, pad, dil)), C, Co8, kh,
                                                             kw, stride, pad, dil, _stream()))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            # weight gradient and the bias gradient in one pass over grad_output
            want_b = has_bias and ctx.needs_input_grad[2]
            gw, gb, acc = _param_grad_buffers(w, ctx.bias_ref if want_b else None, ctx.needs_input_grad[1], want_b)
            _lib.check(lib.lsn_conv2d_backward_weight(_p(x), _p(go), _p(gw), _p(gb), B, H, W, C, Co, kh, kw, stride, pad,
                                                      dil, acc, _stream()))
            gw, gb = _param_grad_results(w, ctx.bias_ref if want_b else None, gw, gb, acc, ctx.needs_input_grad[1])
        return gx, gw, gb, None, None, None, None


class _GroupConvFn(torch.autograd.Function):
    """Grouped convolution (ResNeXt): forward, data gradient, weight / bias gradient through csrc/gconv.hip."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil, groups):
        lib = _lib.load()
        B, C, H, W = x.shape
        Co, _, kh, kw = w.shape
        Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
        w = w.contiguous(memory_format=_CL)
        out = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
        _lib.check(lib.lsn_grouped_conv2d_forward(_p(x), _p(w), _p(bias), _p(out), B, H, W, C, Co, kh, kw, stride, pad, dil,
                                                  groups, 0, _stream()))
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, groups, bias is not None)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        stride, pad, dil, groups, has_bias = ctx.cfg
        lib = _lib.load()
        go = go.contiguous(memory_format=_CL)
        B, C, H, W = x.shape
        Co, _, kh, kw = w.shape
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x, memory_format=_CL)
            _lib.check(lib.lsn_grouped_conv2d_backward_data(_p(go), _p(w), _p(gx), B, H, W, C, Co, kh, kw, stride, pad, dil,
                                                            groups, _stream()))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(w)
            gb = torch.empty(Co, device=x.device, dtype=torch.float32) if has_bias and ctx.needs_input_grad[2] else None
            _lib.check(lib.lsn_grouped_conv2d_backward_weight(_p(x), _p(go), _p(gw), _p(gb), B, H, W, C, Co, kh, kw, stride,
                                                              pad, dil, groups, 0, _stream()))
            if not ctx.needs_input_grad[1]:
                gw = None
        return gx, gw, gb, None, None, None, None


def hip_group_conv_ok(x, weight, stride, padding, dilation, groups, padding_mode='zeros'):
    """Shapes the grouped kernels take (include/lsnet_hip.h): in = out = 4, 8, 16 or 32 channels per group, <= 9 taps."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=_CL)):
        return False
    if groups <= 1 or padding_mode != 'zeros' or isinstance(padding, str):
        return False
    if stride[0] != stride[1] or padding[0] != padding[1] or dilation[0] != dilation[1]:
        return False
    Co, cg, kh, kw = weight.shape
    if x.shape[1] != cg * groups or Co != cg * groups or cg not in (4, 8, 16, 32) or kh * kw > 9:
        return False
    return x.numel() < 2 ** 31 and Co * x.shape[0] * x.shape[2] * x.shape[3] < 2 ** 31


def conv_multi_fwd(xs, w, bias, pad, dil, relu, residuals=None):
    """[conv2d(x, w, bias, 1, pad, dil) for x in xs] in ONE launch (channels-last device tensors; no autograd).
    residuals: per-map tensors of the outputs' shape, added in the epilogue (before the ReLU, lsn_conv_level.residual)."""
    Co, C, kh, kw = w.shape
    n = len(xs)
    levels = (_lib.ConvLevel * n)()
    outs = []
    for i, x in enumerate(xs):
        B, _, H, W = x.shape
        Ho, Wo = H + 2 * pad - (dil * (kh - 1) + 1) + 1, W + 2 * pad - (dil * (kw - 1) + 1) + 1
        out = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
        outs.append(out)
        L = levels[i]
        L.x, L.out, L.B, L.H, L.W = _p(x), _p(out), B, H, W
        if residuals is not None:
            assert residuals[i].shape == out.shape and residuals[i].is_contiguous(memory_format=_CL)
            L.residual = _p(residuals[i])
    _lib.check(_lib.load().lsn_conv2d_forward_prepared(n, levels, _p(weight_image(w, 0)), _p(bias), C, C, Co, kh, kw, 1, pad, dil,
                                                       1 if relu else 0, _stream()))
    return outs


def conv_multi_dgrad(gos, w, xs, pad, dil, accumulate_into=None):
    """Data gradients of conv_multi_fwd for every map in ONE launch.  accumulate_into: per-map tensors of the inputs' shape
    that already hold another path's gradient of the same input -- the results are ADDED to them in the epilogue of the
    launch (lsn_conv_level.residual aliasing out) and they are returned."""
    Co, C, kh, kw = w.shape
    n = len(xs)
    w8, Co8, gos8 = w, Co, gos
    if Co % 4 and C <= 64:    # (narrow tiles read grad_output in 4-channel pieces: zero filters are free)
        Co8 = (Co + 3) // 4 * 4
        gos8 = [_pad_channels(g, Co8) for g in gos]
        w8 = w.new_zeros((Co8, C, kh, kw)).contiguous(memory_format=_CL)
        w8[:Co] = w
    levels = (_lib.ConvLevel * n)()
    gxs = []
    for i, x in enumerate(xs):
        gx = accumulate_into[i] if accumulate_into is not None else torch.empty_like(x, memory_format=_CL)
        gxs.append(gx)
        L = levels[i]
        L.x, L.out, L.B, L.H, L.W = _p(gos8[i]), _p(gx), x.shape[0], x.shape[2], x.shape[3]
        if accumulate_into is not None:
            L.residual = _p(gx)
    _lib.check(_lib.load().lsn_conv2d_backward_data_prepared(n, levels, _p(weight_image(w8, 1, 1, pad, dil)), C, Co8, kh, kw, 1,
                                                             pad, dil, _stream()))
    return gxs


def conv_multi_wgrad(xs, gos, w, bias, pad, dil, want_w=True, want_b=True):
    """Weight / bias gradient summed over the maps in ONE launch; into the parameters' gradient sinks when they have them
    (then (None, None) goes back to autograd)."""
    Co, C, kh, kw = w.shape
    n = len(xs)
    levels = (_lib.ConvLevel * n)()
    for i, x in enumerate(xs):
        L = levels[i]
        L.x, L.grad_out, L.B, L.H, L.W = _p(x), _p(gos[i]), x.shape[0], x.shape[2], x.shape[3]
    want_b = want_b and bias is not None
    gw, gb, acc = _param_grad_buffers(w, bias if want_b else None, want_w, want_b)
    _lib.check(_lib.load().lsn_conv2d_backward_weight_multi(n, levels, _p(gw), _p(gb), C, Co, kh, kw, 1, pad, dil, acc, _stream()))
    return _param_grad_results(w, bias if want_b else None, gw, gb, acc, want_w)


class _ConvMultiFn(torch.autograd.Function):
    """Several input maps of different sizes under ONE weight (the FPN levels of LSHead's shared convolutions):
    forward, data gradients and the weight / bias gradient (summed over the maps) in one launch each.  Stride 1."""

    @staticmethod
    def forward(ctx, w, bias, cfg, *xs):
        pad, dil, relu = cfg[:3]
        nres = cfg[3] if len(cfg) > 3 else 0     # the last nres tensors are residuals of the first maps' outputs (round 5)
        res = list(xs[len(xs) - nres:]) if nres else None
        xs = xs[:len(xs) - nres] if nres else xs
        w = w.contiguous(memory_format=_CL)
        outs = conv_multi_fwd(xs, w, bias, pad, dil, relu, res)
        ctx.save_for_backward(w, *xs, *(outs if relu else []))
        ctx.cfg, ctx.n, ctx.has_bias, ctx.nres = (pad, dil, relu), len(xs), bias is not None, nres
        ctx.bias_ref = bias
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gos):
        pad, dil, relu = ctx.cfg
        n = ctx.n
        saved = ctx.saved_tensors
        w, xs = saved[0], saved[1:1 + n]
        outs = saved[1 + n:] if relu else None
        if relu and GATE_MULTI:
            gos = relu_gate_multi(list(gos), outs)      # one launch for all maps, reading sliced gradients where they lie
        elif relu:
            gos = [g.contiguous(memory_format=_CL) for g in gos]
            gos = [relu_gate(g, o) if o.numel() % 4 == 0 else g * (o > 0) for g, o in zip(gos, outs)]
        else:
            gos = [g.contiguous(memory_format=_CL) for g in gos]
        need_x = [ctx.needs_input_grad[3 + i] for i in range(n)]
        gxs = [None] * n
        if any(need_x):
            gxs = [g if need else None for g, need in zip(conv_multi_dgrad(gos, w, xs, pad, dil), need_x)]
        gw = gb = None
        if ctx.needs_input_grad[0] or (ctx.has_bias and ctx.needs_input_grad[1]):
            gw, gb = conv_multi_wgrad(xs, gos, w, ctx.bias_ref, pad, dil, ctx.needs_input_grad[0],
                                      ctx.has_bias and ctx.needs_input_grad[1])
        # the residual sits in front of the ReLU: its gradient is the (gated) output gradient itself
        gres = [gos[i] if ctx.needs_input_grad[3 + n + i] else None for i in range(ctx.nres)]
        return (gw, gb, None, *gxs, *gres)


def conv2d_multi(xs, weight, bias=None, padding=0, dilation=1, relu=False, residuals=None):
    """[conv2d(x, weight, bias, 1, padding, dilation) (+ residual) for x in xs] in one launch per pass (stride 1, up to 8 maps).
    residuals: one tensor of the output's shape per map, added in the launch's epilogue (before the ReLU) -- the sum of two
    branches without an element-wise launch of its own (LSHead's `relu(1x1(gathered)) + 3x3(tower)`, lsnet_head.py:640-755)."""
    nres = 0 if residuals is None else len(residuals)
    assert nres in (0, len(xs))
    return list(_ConvMultiFn.apply(weight, bias, (int(padding), int(dilation), bool(relu), nres), *xs, *(residuals or [])))


def hip_conv_ok(x, weight, stride, padding, dilation, groups, padding_mode='zeros'):
    """The own kernels take every dense (groups = 1) fp32 convolution on the device, in every math mode and memory
    format: a contiguous (NCHW) tensor is re-laid channels-last on the way in (`_as_cl`, a layout copy -- the result stays
    channels-last, as ATen's own convolution propagates the format), and LSN_MATH_FP32 runs the fp32-EQUIVALENT 6-term
    kernels here (exactness of that mode concerns the deformable family; there is no fp32-MFMA dense kernel, and no vendor
    convolution is called either).  What is left for F.conv2d: other dtypes / devices, 'same' / non-zero padding modes,
    unequal strides, tensors beyond 32-bit byte offsets."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        return False
    if groups != 1 or padding_mode != 'zeros' or isinstance(padding, str):
        return False
    if stride[0] != stride[1] or padding[0] != padding[1] or dilation[0] != dilation[1]:
        return False
    if x.numel() * 4 >= 2 ** 31 or weight.numel() * 8 >= 2 ** 31 or weight.shape[2] * weight.shape[3] > 64:
        return False
    # the output (and with it grad_output of the backward passes) must fit 32-bit byte offsets too: csrc/conv.hip conv_check
    kh, kw = weight.shape[2], weight.shape[3]
    ho = (x.shape[2] + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    wo = (x.shape[3] + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    if ho <= 0 or wo <= 0 or x.shape[0] * ho * wo * weight.shape[0] * 4 >= 2 ** 31:
        return False
    return True


def _as_cl(x):
    return x if x.is_contiguous(memory_format=_CL) else x.contiguous(memory_format=_CL)


def _stem_forward(x, weight, bias, stride, pad, relu):
    """Forward of a shallow wide-kernel convolution (the 7x7 stem on the 3-channel image) in the row-merged form of
    lsn_conv2d_forward_pitched: the image is zero-padded (space: `pad`, channels: to 4) once, then a tap row of kw
    pixels x 4 channels is ONE contiguous 4 kw-float "channel" run -- kh chunks of the implicit GEMM instead of kh * kw
    chunks that are 1/8 full.  No gradient (the stem is frozen in every LSNet config)."""
    lib = _lib.load()
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    C4 = (C + 3) // 4 * 4
    xp = x.new_zeros((B, C4, H + 2 * pad, W + 2 * pad)).contiguous(memory_format=_CL)
    xp[:, :C, pad:pad + H, pad:pad + W] = x
    w4 = weight.new_zeros((Co, C4, kh, kw)).contiguous(memory_format=_CL)     # memory (Co, kh, kw, C4) = (Co, kh, 1, kw C4)
    w4[:, :C] = weight
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
    _lib.check(lib.lsn_conv2d_forward_pitched(_p(xp), _p(w4), _p(bias), _p(out), None, B, H + 2 * pad, W + 2 * pad,
                                              kw * C4, C4, Co, kh, 1, stride, 0, 1, 1 if relu else 0, _stream()))
    return out


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False):
    C = x.shape[1]
    no_grad = not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or
                                                (bias is not None and bias.requires_grad)))
    if C < 8 and weight.shape[3] > 1 and int(dilation) == 1 and no_grad:
        return _stem_forward(x, weight, bias, int(stride), int(padding), bool(relu))
    if C % 4:    # the 3-channel image of the stem: one zero channel (and a zero column of taps)
        C4 = (C + 3) // 4 * 4
        x = _pad_channels(x, C4)
        w4 = weight.new_zeros((weight.shape[0], C4, weight.shape[2], weight.shape[3])).contiguous(memory_format=_CL)
        w4[:, :C] = weight
        weight = w4
    return _ConvFn.apply(x, weight, bias, int(stride), int(padding), int(dilation), bool(relu))


_folded = {}    # (id(conv.weight), id(bn)) -> [weakrefs, versions, folded weight, shift]


def _fold_bn(conv, bn):
    """(w * scale[co], shift) with scale = gamma / sqrt(var + eps), shift = beta - mean * scale: a frozen BatchNorm behind
    a frozen convolution is the same convolution with other constants.  Rebuilt only when a tensor's version moves."""
    key = (id(conv.weight), id(bn))
    vers = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
            conv.weight.data_ptr())
    ent = _folded.get(key)
    if ent is not None and ent[0]() is conv.weight and ent[1]() is bn and ent[2] == vers:
        return ent[3], ent[4]
    with torch.no_grad():
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = (bn.bias - bn.running_mean * scale).contiguous()
        w = (conv.weight * scale.view(-1, 1, 1, 1)).contiguous(memory_format=_CL)
    _folded[key] = [weakref.ref(conv.weight, lambda _, k=key: _folded.pop(k, None)), weakref.ref(bn), vers, w, shift]
    return w, shift


def conv_bn_act_frozen(conv, bn, x, relu=True, residual=None):
    """relu(bn(conv(x)) + residual) in ONE launch for a FROZEN conv + eval-mode BatchNorm pair whose input needs no
    gradient (stem and stage 1 of every LSNet backbone, `frozen_stages=1`: resnet.py:569-585, 636-645): the norm is folded
    into the weight / bias, residual add and ReLU ride in the convolution's epilogue (include/lsnet_hip.h
    lsn_conv_level.residual) -- no normalisation pass over the largest activations of the network.  Returns None when the
    pair does not qualify (the caller then runs conv and bn_act separately)."""
    if not isinstance(conv, Conv2d) or conv.bias is not None or conv.groups != 1 or bn.training or not bn.affine \
            or not bn.track_running_stats:
        return None
    if conv.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad \
            or (torch.is_grad_enabled() and x.requires_grad):
        return None
    if not hip_conv_ok(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups, conv.padding_mode):
        return None
    x = _as_cl(x)
    w, shift = _fold_bn(conv, bn)
    stride, pad, dil = conv.stride[0], conv.padding[0], conv.dilation[0]
    B, C, H, W = x.shape
    Co, _, kh, kw = w.shape
    if C < 8:    # the image stem: row-merged form, no residual there
        if residual is not None or kw == 1 or dil != 1:
            return None
        return _stem_forward(x, w, shift, stride, pad, relu)
    if C % 4:
        return None
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    if residual is not None:
        if not (tuple(residual.shape) == (B, Co, Ho, Wo) and residual.dtype == torch.float32 and Co % 4 == 0):
            return None
        residual = _as_cl(residual)
    lib = _lib.load()
    out = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL)
    lv = _levels(1)
    lv[0].x, lv[0].out, lv[0].B, lv[0].H, lv[0].W, lv[0].residual = _p(x), _p(out), B, H, W, _p(residual)
    _lib.check(lib.lsn_conv2d_forward_prepared(1, lv, _p(weight_image(w, 0)), _p(shift), C, C, Co, kh, kw, stride, pad,
                                               dil, 1 if relu else 0, _stream()))
    return out


class _ConvBnActFn(torch.autograd.Function):
    """y = act(bn(conv(x)) + residual) with the eval-mode BatchNorm FOLDED into the convolution (one launch, the raw
    convolution output is never stored) and every gradient -- input, residual, weight, gamma, beta -- from the library.
    Backward: g = ReLU gate of grad_y (one streaming pass; g is also the residual's gradient); the data gradient runs
    on the backward image of the SCALED weight; lsn_conv2d_backward_weight_bn forms grad_w, grad_gamma and grad_beta from
    ONE weight-gradient launch with conv = w . x pulled out of the pixel sum -- exact for every gamma (a zero-initialised
    norm3, `zero_init_residual`, learns; nothing is divided by gamma)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, residual, bn, stride, pad, dil, relu):
        out = conv_fwd_bn(x, w, bn, stride, pad, dil, relu, residual)
        ctx.save_for_backward(x, w, out if relu else None)
        ctx.cfg = (stride, pad, dil, relu, residual is not None)
        ctx.bn = bn
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, dil, relu, has_res = ctx.cfg
        bn = ctx.bn
        gy = gy.contiguous(memory_format=_CL)
        need_x, need_w, need_g, need_b = ctx.needs_input_grad[:4]
        need_res = has_res and ctx.needs_input_grad[4]
        g = relu_gate(gy, y) if relu else gy
        gx = gw = dg = db = None
        if need_x:
            gx = dgrad(g, w, x.shape, stride, pad, dil, bn=bn)
        if need_w or need_g or need_b:
            gw, dg, db = wgrad_bn(x, g, w, bn, stride, pad, dil, (need_w, need_g, need_b))
        return gx, gw, dg, db, (g if need_res else None), None, None, None, None, None


def conv_bn_act(conv, bn, x, relu=True, residual=None):
    """relu(bn(conv(x)) + residual) in ONE forward launch for a TRAINABLE convolution + eval-mode BatchNorm pair (stages
    2 - 4 of the LSNet backbones: `norm_eval=True` keeps the statistics fixed while gamma / beta train,
    resnet.py:636-645): the norm's scale is folded into the weight image that is rebuilt after every optimizer step
    anyway, its shift rides in the bias slot, residual add and ReLU in the epilogue.  Saves the normalisation pass over
    every backbone activation (a read and a write) and the storage of the raw convolution output.  Returns None when the
    pair does not qualify (the caller then runs conv and bn_act separately)."""
    if not isinstance(conv, Conv2d) or conv.bias is not None or conv.groups != 1 or bn.training or not bn.affine \
            or not bn.track_running_stats or not isinstance(bn, nn.BatchNorm2d):
        return None
    if not hip_conv_ok(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups, conv.padding_mode):
        return None
    Co, C, kh, kw = conv.weight.shape
    if C % 4 or Co % 4 or not (256 % (Co // 4) == 0 if Co <= 1024 else Co % 1024 == 0):
        return None
    if not conv.weight.is_contiguous(memory_format=_CL):
        return None
    x = _as_cl(x)
    stride, pad, dil = conv.stride[0], conv.padding[0], conv.dilation[0]
    B, _, H, W = x.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    if residual is not None:
        if not (tuple(residual.shape) == (B, Co, Ho, Wo) and residual.dtype == torch.float32):
            return None
        residual = _as_cl(residual)
    return _ConvBnActFn.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, stride, pad, dil, bool(relu))


class Conv2d(nn.Conv2d):

    def forward(self, x, weight=None):
        """`weight`: use this tensor instead of the module's parameter (a zero-padded view of it, for inputs whose
        channel count was rounded up to the kernels' granularity)."""
        if weight is not None:
            return self._run(x, weight)
        return self._run(x, self.weight)

    def forward_multi(self, xs, relu=False, residuals=None):
        """The same convolution over several maps (FPN levels) in ONE launch per pass.  residuals: per-map tensors added to the
        outputs (before the ReLU) -- in the launch's epilogue on the device."""
        xs = list(xs)
        if (1 < len(xs) <= 8 and self.stride[0] == 1 and xs[0].shape[1] % 4 == 0
                and all(hip_conv_ok(x, self.weight, self.stride, self.padding, self.dilation, self.groups,
                                    self.padding_mode) for x in xs)
                and (residuals is None or (self.weight.shape[0] % 4 == 0 and all(r.is_cuda and r.dtype == torch.float32 for r in residuals)))):
            res = None if residuals is None else [_as_cl(r) for r in residuals]
            return conv2d_multi([_as_cl(x) for x in xs], self.weight, self.bias, self.padding[0], self.dilation[0], relu, res)
        outs = [self._run(x, self.weight) for x in xs]
        if residuals is not None:
            outs = [o + r for o, r in zip(outs, residuals)]
        return [F.relu(o) for o in outs] if relu else outs

    def _run(self, x, w):
        if hip_conv_ok(x, w, self.stride, self.padding, self.dilation, self.groups, self.padding_mode):
            return conv2d(_as_cl(x), w, self.bias, self.stride[0], self.padding[0], self.dilation[0], False)
        if hip_group_conv_ok(x, w, self.stride, self.padding, self.dilation, self.groups, self.padding_mode):
            return _GroupConvFn.apply(x, w, self.bias, self.stride[0], self.padding[0], self.dilation[0], self.groups)
        _warn_aten_fallback('Conv2d', x, f'weight {tuple(w.shape)}, stride {self.stride}, groups {self.groups}, '
                                           f'padding_mode {self.padding_mode}, input {tuple(x.shape)} {x.dtype}')
        return F.conv2d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)
