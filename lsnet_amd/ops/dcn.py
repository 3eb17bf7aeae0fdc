"""Deformable convolution family -- Python mirror of mmdet/ops/dcn/deform_conv.py.

Same public names, argument meaning and error behaviour as the reference
(`deform_conv`, `modulated_deform_conv`, `pyramid_deform_conv`, `DeformConv`, `DeformConvPack`,
`ModulatedDeformConv`, `ModulatedDeformConvPack`, `PyramidDeformConv`;
deform_conv.py:15-111,114-185,188-287,295-630), plus `dcn_multi`: several (input, offset, mask)
triples that share one weight -- e.g. the five FPN levels of a shared tower -- in ONE launch.

All of them funnel into one autograd Function that calls the C ABI (lsn_dcn_forward /
lsn_dcn_backward, include/lsnet_hip.h).  Tensors stay logically NCHW; when they are in
channels-last memory the kernels run without any layout copy.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from ..cnn.registry import CONV_LAYERS
from . import grad_sink
from .backend import get_backend
from . import conv as _conv
from .conv import Conv2d


def _conv_out(size, k, stride, pad, dil):
    return (size + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def _same_int(v, what):
    v = _pair(v)
    if v[0] != v[1]:
        raise NotImplementedError(f'{what} must be the same for height and width, got {v}')
    return int(v[0])


class _DCNFunction(Function):
    """forward(weight, bias, cfg, n, x_0..x_{n-1}, off_0..off_{n-1}, mask_0..mask_{n-1})"""

    @staticmethod
    def forward(ctx, weight, bias, cfg, n, *tensors):
        inputs, offsets, masks = tensors[:n], tensors[n:2 * n], tensors[2 * n:3 * n]
        backend = get_backend(inputs[0])
        out_hw = []
        for x, off in zip(inputs, offsets):
            if cfg['pyramid']:
                hw = (_conv_out(off.shape[2], weight.shape[2], cfg['stride'], cfg['pad'], cfg['dil']),
                      _conv_out(off.shape[3], weight.shape[3], cfg['stride'], cfg['pad'], cfg['dil']))
            else:
                hw = (_conv_out(x.shape[2], weight.shape[2], cfg['stride'], cfg['pad'], cfg['dil']),
                      _conv_out(x.shape[3], weight.shape[3], cfg['stride'], cfg['pad'], cfg['dil']))
            if not all(s > 0 for s in hw):
                raise ValueError('convolution input is too small (output would be '
                                 f'{x.shape[0]}x{weight.shape[0]}x{hw[0]}x{hw[1]})')
            if tuple(off.shape[2:]) != hw:
                raise RuntimeError(f'invalid spatial size of offset, expected height: {hw[0]} width: {hw[1]}, '
                                   f'but got height: {off.shape[2]} width: {off.shape[3]}')
            if off.shape[1] != cfg['dg'] * (3 if cfg.get('fused_om') else 2) * weight.shape[2] * weight.shape[3]:
                raise RuntimeError('invalid number of channels of offset')
            out_hw.append(hw)
        ctx.cfg, ctx.n, ctx.backend = cfg, n, backend
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias    # backward only asks where its gradient goes (ops/grad_sink.py)
        ctx.mask_none = [m is None for m in masks]
        ctx.save_for_backward(weight, *inputs, *offsets, *[m for m in masks if m is not None])
        outs = backend.dcn_forward(list(inputs), list(offsets), list(masks), weight, bias, cfg, out_hw)
        cat = int(cfg.get('concat') or 0)
        if cat > 1 and len(outs) == n:   # a backend that left the levels apart
            outs = [torch.cat(outs[j:j + cat], dim=1) for j in range(0, n, cat)]
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grad_outs):
        n, cfg = ctx.n, ctx.cfg
        saved = ctx.saved_tensors
        weight, inputs, offsets = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n]
        rest = list(saved[1 + 2 * n:])
        masks = [None if none else rest.pop(0) for none in ctx.mask_none]
        nig = ctx.needs_input_grad  # (weight, bias, cfg, n, *tensors)
        need = dict(weight=nig[0], bias=ctx.has_bias and nig[1],
                    input=[nig[4 + i] for i in range(n)],
                    offset=[nig[4 + n + i] for i in range(n)],
                    mask=[nig[4 + 2 * n + i] for i in range(n)])
        gos = []
        cat = int(cfg.get('concat') or 0)
        if cat > 1:   # one gradient per group of `cat` levels: the levels read their channel slices of it
            Co = weight.shape[0]
            for j, g in enumerate(grad_outs):
                x, off = inputs[j * cat], offsets[j * cat]
                if g is None:
                    g = x.new_zeros((x.shape[0], cat * Co, off.shape[2], off.shape[3]))
                if x.is_contiguous(memory_format=torch.channels_last):
                    g = g.contiguous(memory_format=torch.channels_last)
                gos += [g.narrow(1, q * Co, Co) for q in range(cat)]
        for i, g in enumerate(grad_outs if cat <= 1 else ()):
            if g is None:  # this level's output was not used downstream
                x, off = inputs[i], offsets[i]
                g = x.new_zeros((x.shape[0], weight.shape[0], off.shape[2], off.shape[3]))
            gos.append(g)
        # parameter gradients straight into their sinks (the all-reduce buckets) when both have one
        sw = grad_sink.sink(weight) if need['weight'] else None
        sb = grad_sink.sink(ctx.bias_ref) if need['bias'] else None
        sunk = sw is not None and (not need['bias'] or sb is not None) and sw.stride() == weight.stride() \
            and getattr(ctx.backend, 'supports_grad_sinks', False)
        if sunk:
            need['sinks'] = (sw, sb)
        gxs, goffs, gmsks, gw, gb = ctx.backend.dcn_backward(list(inputs), list(offsets), masks, weight, gos,
                                                            cfg, need)
        if sunk and need.get('sunk'):
            grad_sink.done(weight)
            if sb is not None:
                grad_sink.done(ctx.bias_ref)
            gw = gb = None
        gxs = [g if need['input'][i] else None for i, g in enumerate(gxs)]
        goffs = [g if need['offset'][i] else None for i, g in enumerate(goffs)]
        gmsks = [g if (need['mask'][i] and not ctx.mask_none[i]) else None for i, g in enumerate(gmsks)]
        if gw is not None and gw.stride() != weight.stride():
            gw = gw.contiguous(memory_format=torch.channels_last if weight.is_contiguous(
                memory_format=torch.channels_last) and not weight.is_contiguous() else torch.contiguous_format)
        return (gw if need['weight'] else None, gb if (need['bias'] and gb is not None) else None, None, None,
                *gxs, *goffs, *gmsks)


class _DCNPackFn(Function):
    """A DCNv2 pack over several maps (the FPN levels of a shared LSHead tower layer) as ONE autograd node:
    offsets | mask logits = conv_offset(x) (one dense launch), out = DCNv2(x, offsets, sigmoid(logits)) (one launch).
    The reference runs them as two operators (deform_conv.py:527-534), and so did this package until round 4 -- with the
    input's two gradients (through the sampling and through conv_offset) meeting in an autograd accumulation per level.
    Here conv_offset's backward-data launch ADDS onto the deformable op's grad_input in its epilogue
    (ops/conv.py conv_multi_dgrad accumulate_into): 30 elementwise adds and as many gradient tensors less per step
    (same-box A/B: 36.30 vs 36.43 ms per step).
    forward(w, b, wo, bo, cfg, n, x_0 .. x_{n-1})"""

    @staticmethod
    def forward(ctx, weight, bias, w_off, b_off, cfg, n, *xs):
        backend = get_backend(xs[0])
        pad, dil = cfg['pad'], cfg['dil']
        w_off = w_off.contiguous(memory_format=torch.channels_last)
        offs = _conv.conv_multi_fwd(xs, w_off, b_off, pad, dil, False)
        out_hw = [tuple(o.shape[2:]) for o in offs]
        outs = backend.dcn_forward(list(xs), offs, [None] * n, weight, bias, cfg, out_hw)
        ctx.cfg, ctx.n, ctx.backend = cfg, n, backend
        ctx.bias_ref, ctx.b_off_ref = bias, b_off
        ctx.save_for_backward(weight, w_off, *xs, *offs)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grad_outs):
        n, cfg = ctx.n, ctx.cfg
        saved = ctx.saved_tensors
        weight, w_off, xs, offs = saved[0], saved[1], saved[2:2 + n], saved[2 + n:2 + 2 * n]
        nig = ctx.needs_input_grad   # (weight, bias, w_off, b_off, cfg, n, *xs)
        need_x = [nig[6 + i] for i in range(n)]
        need_off = nig[2] or nig[3] or any(need_x)
        need = dict(weight=nig[0], bias=ctx.bias_ref is not None and nig[1], input=[True] * n if any(need_x) else [False] * n,
                    offset=[need_off] * n, mask=[False] * n)
        gos = [g if g is not None else x.new_zeros((x.shape[0], weight.shape[0], o.shape[2], o.shape[3]))
               for g, x, o in zip(grad_outs, xs, offs)]
        sw = grad_sink.sink(weight) if need['weight'] else None
        sb = grad_sink.sink(ctx.bias_ref) if need['bias'] else None
        sunk = sw is not None and (not need['bias'] or sb is not None) and sw.stride() == weight.stride() \
            and getattr(ctx.backend, 'supports_grad_sinks', False)
        if sunk:
            need['sinks'] = (sw, sb)
        gxs, goffs, _, gw, gb = ctx.backend.dcn_backward(list(xs), list(offs), [None] * n, weight, gos, cfg, need)
        if sunk and need.get('sunk'):
            grad_sink.done(weight)
            if sb is not None:
                grad_sink.done(ctx.bias_ref)
            gw = gb = None
        pad, dil = cfg['pad'], cfg['dil']
        if any(need_x):   # conv_offset's data gradient lands ON the deformable op's grad_input
            gxs = _conv.conv_multi_dgrad(goffs, w_off, xs, pad, dil, accumulate_into=gxs)
        gwo = gbo = None
        if nig[2] or nig[3]:
            gwo, gbo = _conv.conv_multi_wgrad(xs, goffs, w_off, ctx.b_off_ref, pad, dil, nig[2], nig[3])
        return (gw if need['weight'] else None, gb if (need['bias'] and gb is not None) else None, gwo, gbo, None, None,
                *[g if nx else None for g, nx in zip(gxs, need_x)])


class _OffsetChainFn(Function):
    """forward(mults, copies, off_0 .. off_{n-1}) -> 3 n copies tensors: for level l the fields off_l m1, (off_l m1) m2,
    ((off_l m1) m2) m3 with m_k = mults[l][k] = (scale_h, scale_w) on the (y, x) channel pairs -- one launch forward, one
    backward.  copies = 2: every field comes back twice (the second an alias), one per consumer, and the backward launch adds
    the two gradients itself -- what autograd's accumulation would have been a launch per field for."""

    @staticmethod
    def forward(ctx, mults, copies, *offs):
        be = get_backend(offs[0])
        ctx.mults, ctx.be, ctx.device, ctx.copies = mults, be, offs[0].device, copies
        ctx.shapes = [tuple(o.shape) for o in offs]
        res = be.offset_chain(list(offs), mults)
        out = [t for trio in res for t in trio]
        if copies == 2:
            out += [t.view(t.shape) for t in out]
        return tuple(out)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        n = len(ctx.shapes)
        gs = [list(grads[3 * l:3 * l + 3]) + (list(grads[3 * n + 3 * l:3 * n + 3 * l + 3]) if ctx.copies == 2 else [])
              for l in range(n)]
        return (None, None, *ctx.be.offset_chain_backward(ctx.shapes, ctx.device, ctx.mults, gs))


def offset_scale_chain(offs, mults, copies=1):
    """LSHead's cumulative offset rescaling (lsnet_head.py:622-638; the in-place `*=` of the reference accumulates over the
    three source levels of a destination level).  offs: per-level (B, 2 taps, H, W) fields; mults[l]: three (scale_h, scale_w)
    pairs.  Returns `copies` lists, each per level the three fields [off m1, off m1 m2, off m1 m2 m3] -- each product rounded
    on its own, as the three multiplications of the reference are.  copies = 2 hands a second consumer its own handles of
    the same fields (on the device their two gradients then meet inside the one backward launch)."""
    offs = list(offs)
    if offs and all(o.is_cuda for o in offs) and len(offs) <= 8 and all(len(m) == 3 for m in mults) and copies in (1, 2):
        be = get_backend(offs[0])
        if all(be.offset_chain_ok(o) for o in offs):
            n = len(offs)
            flat = _OffsetChainFn.apply(tuple(tuple((float(a), float(b)) for a, b in m) for m in mults), copies, *offs)
            return [[list(flat[c * 3 * n + 3 * l:c * 3 * n + 3 * l + 3]) for l in range(n)] for c in range(copies)]
    res = []
    for off, m in zip(offs, mults):
        cur, trio = off, []
        for sh, sw in m:
            mult = _scale_const(float(sh), float(sw), off.shape[1], off.device, off.dtype)
            cur = cur * mult
            trio.append(cur)
        res.append(trio)
    return [res] * copies


_SCALE_CONSTS = {}


def _scale_const(sh, sw, channels, device, dtype):
    """The (1, C, 1, 1) multiplier [sh, sw, sh, sw, ...] of the fallback above, cached: `new_tensor` on a device tensor is a
    synchronous host-to-device copy in the middle of the step, and breaks a stream capture."""
    key = (sh, sw, channels, str(device), dtype)
    t = _SCALE_CONSTS.get(key)
    if t is None:
        t = torch.tensor([sh, sw], dtype=dtype).repeat(channels // 2).view(1, -1, 1, 1).to(device)
        _SCALE_CONSTS[key] = t
    return t


def dcn_multi(inputs, offsets, masks, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
              deformable_groups=1, scales=None, pyramid=False, fused_om=False, concat=0):
    """Batched deformable convolution: out_i = DCN(inputs[i], offsets[i], masks[i]; weight, bias).

    masks may be None (DCNv1 / pyramid) or a list with None entries.  scales: per-level
    (scale_h, scale_w) for the pyramid op (default 1).  With pyramid=True the output grid is the
    offset grid (deform_conv.py:215-217).

    fused_om=True: `offsets[i]` is the raw (B, 3*dg*kh*kw, H, W) output of a DCNv2 pack's conv_offset
    (offsets in the first two thirds of the channels, mask LOGITS in the last third); the sigmoid and
    its derivative are applied inside the kernels and ONE gradient tensor comes back.

    concat=g > 1: returns len(inputs) / g tensors, torch.cat(out[j g : (j + 1) g], dim=1) -- the outputs of g consecutive
    levels (same batch and output size) side by side, written there by the kernels where they can."""
    n = len(inputs)
    if masks is None:
        masks = [None] * n
    if scales is None:
        scales = [(1.0, 1.0)] * n
    for x in inputs:
        if x is not None and x.dim() != 4:
            raise ValueError(f'Expected 4D tensor as input, got {x.dim()}D tensor instead.')
    cfg = dict(stride=_same_int(stride, 'stride'), pad=_same_int(padding, 'padding'),
               dil=_same_int(dilation, 'dilation'), groups=int(groups), dg=int(deformable_groups),
               scales=[(float(a), float(b)) for a, b in scales], pyramid=bool(pyramid), fused_om=bool(fused_om),
               concat=int(concat) if concat and concat > 1 else 0)
    return list(_DCNFunction.apply(weight, bias, cfg, n, *inputs, *offsets, *masks))


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
    """DCNv1, deform_conv.py:15-111 (`DeformConvFunction.apply`)."""
    if input is not None and input.dim() != 4:
        raise ValueError(f'Expected 4D tensor as input, got {input.dim()}D tensor instead.')
    cur = min(im2col_step, input.shape[0])
    assert (input.shape[0] % cur) == 0, 'im2col step must divide batchsize'
    return dcn_multi([input], [offset], None, weight, None, stride, padding, dilation, groups,
                     deformable_groups)[0]


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                          deformable_groups=1):
    """DCNv2, deform_conv.py:114-185 (`ModulatedDeformConvFunction.apply`)."""
    return dcn_multi([input], [offset], [mask], weight, bias, stride, padding, dilation, groups,
                     deformable_groups)[0]


def pyramid_deform_conv(input, offset, weight, scales=1, stride=1, padding=0, dilation=1, groups=1,
                        deformable_groups=1, im2col_step=64):
    """Pyramid DCN, deform_conv.py:188-287: DCNv1 sampling a map of another resolution; `scales`
    = (scale_h, scale_w) multiply the regular grid position before the offset is added."""
    if input is not None and input.dim() != 4:
        raise ValueError(f'Expected 4D tensor as input, got {input.dim()}D tensor instead.')
    cur = min(im2col_step, input.shape[0])
    assert (input.shape[0] % cur) == 0, 'im2col step must divide batchsize'
    return dcn_multi([input], [offset], None, weight, None, stride, padding, dilation, groups,
                     deformable_groups, scales=[_pair(scales)], pyramid=True)[0]


class DeformConv(nn.Module):
    """deform_conv.py:295-357"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, f'in_channels {in_channels} is not divisible by groups {groups}'
        assert out_channels % groups == 0, f'out_channels {out_channels} is not divisible by groups {groups}'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed = False  # nn.Conv2d compatibility
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        # pad inputs smaller than the kernel (deform_conv.py:339-356)
        pad_h = max(self.kernel_size[0] - x.size(2), 0)
        pad_w = max(self.kernel_size[1] - x.size(3), 0)
        if pad_h or pad_w:
            x = F.pad(x, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
            offset = F.pad(offset, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                          self.deformable_groups)
        if pad_h or pad_w:
            out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
        return out


@CONV_LAYERS.register_module(name='DCN')
class DeformConvPack(DeformConv):
    """DCNv1 with its own offset conv, deform_conv.py:360-435."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = Conv2d(self.in_channels,
                                     self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding), dilation=_pair(self.dilation), bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        offset = self.conv_offset(x)
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        _remap_old_offset_keys(state_dict, prefix, local_metadata)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)


def _remap_old_offset_keys(state_dict, prefix, local_metadata):
    """Checkpoints written before version 2 name the offset conv `<name>_offset`
    (deform_conv.py:410-425, 536-552)."""
    version = local_metadata.get('version', None)
    if version is None or version < 2:
        for suffix in ('weight', 'bias'):
            new, old = prefix + 'conv_offset.' + suffix, prefix[:-1] + '_offset.' + suffix
            if new not in state_dict and old in state_dict:
                state_dict[new] = state_dict.pop(old)


class ModulatedDeformConv(nn.Module):
    """deform_conv.py:438-484"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.with_bias = bias
        self.transposed = False
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.init_weights()

    def init_weights(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


@CONV_LAYERS.register_module(name='DCNv2')
class ModulatedDeformConvPack(ModulatedDeformConv):
    """DCNv2 with its own offset/mask conv (zero-initialised), deform_conv.py:488-562.

    `forward_multi` runs the same layer over several feature maps (the FPN levels of a shared
    tower) with one kernel launch for the deformable part."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = Conv2d(self.in_channels,
                                     self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding), dilation=_pair(self.dilation), bias=True)
        self.init_weights()

    def init_weights(self):
        super().init_weights()
        if hasattr(self, 'conv_offset'):
            self.conv_offset.weight.data.zero_()
            self.conv_offset.bias.data.zero_()

    def _offset_mask(self, x):
        out = self.conv_offset(x)
        k2 = 2 * self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        # chunk(3) + cat(o1, o2) of the reference == the first 2/3 of the channels
        return out[:, :k2], torch.sigmoid(out[:, k2:])

    def forward(self, x):
        return self.forward_multi([x])[0]

    def forward_multi(self, xs):
        # conv_offset's output goes to the op as ONE tensor (offsets | mask logits): no chunk / cat /
        # sigmoid kernels and a single dense gradient for conv_offset's backward
        xs = list(xs)
        co = self.conv_offset
        if (1 <= len(xs) <= 8 and _same_int(self.stride, 'stride') == 1 and self.groups == 1 and xs[0].shape[1] % 4 == 0
                and all(x.is_cuda and x.is_contiguous(memory_format=torch.channels_last) and
                        _conv.hip_conv_ok(x, co.weight, co.stride, co.padding, co.dilation, co.groups, co.padding_mode)
                        for x in xs)):
            cfg = dict(stride=1, pad=_same_int(self.padding, 'padding'), dil=_same_int(self.dilation, 'dilation'), groups=1,
                       dg=int(self.deformable_groups), scales=[(1.0, 1.0)] * len(xs), pyramid=False, fused_om=True)
            return list(_DCNPackFn.apply(self.weight, self.bias, co.weight, co.bias, cfg, len(xs), *xs))   # one autograd node
        return dcn_multi(xs, co.forward_multi(xs), None, self.weight, self.bias,
                         self.stride, self.padding, self.dilation, self.groups, self.deformable_groups,
                         fused_om=True)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        _remap_old_offset_keys(state_dict, prefix, local_metadata)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)


class PyramidDeformConv(nn.Module):
    """DCNv1 whose output grid is the offset grid and whose source map has another resolution
    (deform_conv.py:565-630).  `forward_multi` evaluates many (source, offset, scale) triples
    with one launch."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, f'in_channels {in_channels} is not divisible by groups {groups}'
        assert out_channels % groups == 0, f'out_channels {out_channels} is not divisible by groups {groups}'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed = False
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def _pad_small(self, x, offset):
        pad_h = max(self.kernel_size[0] - x.size(2), 0)
        pad_w = max(self.kernel_size[1] - x.size(3), 0)
        if pad_h or pad_w:  # deform_conv.py:614-621
            x = F.pad(x, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
            offset = F.pad(offset, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
        return x, offset, pad_h, pad_w

    def forward(self, x, offset, scale_h, scale_w):
        return self.forward_multi([x], [offset], [(scale_h, scale_w)])[0]

    def forward_multi(self, xs, offsets, scales, weight=None, concat=0):
        """`weight`: a zero-padded view of `self.weight` for sources whose channels were rounded up.
        `concat` = g > 1: the outputs of g consecutive pairs (one destination level) come back as ONE tensor, concatenated
        along the channels (lsnet_head.py:640-647 concatenates them right away)."""
        prepped = [self._pad_small(x, o) for x, o in zip(xs, offsets)]
        if concat and concat > 1 and any(p[2] or p[3] for p in prepped):   # padded tiny sources: per pair, then concatenate
            outs = self.forward_multi(xs, offsets, scales, weight)
            return [torch.cat(outs[j:j + concat], dim=1) for j in range(0, len(outs), concat)]
        outs = dcn_multi([p[0] for p in prepped], [p[1] for p in prepped], None,
                         self.weight if weight is None else weight, None, self.stride,
                         self.padding, self.dilation, self.groups, self.deformable_groups,
                         scales=[_pair(s) for s in scales], pyramid=True, concat=concat)
        if concat and concat > 1:
            return outs
        res = []
        for out, (_, _, pad_h, pad_w) in zip(outs, prepped):
            if pad_h or pad_w:
                out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
            res.append(out)
        return res
