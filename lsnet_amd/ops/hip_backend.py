"""The 'cuda' (HIP) backend: marshals torch tensors into the C ABI of liblsnet_hip.so."""
import ctypes

import torch

from .. import _lib
from .backend import register_backend

_CL = torch.channels_last
# LSNET_CACHE_DCN_IMAGES=0: the deformable layers build their weight images inside every call, as until round 5 (A/B switch)
CACHE_DCN_IMAGES = __import__('os').environ.get('LSNET_CACHE_DCN_IMAGES', '1') != '0'


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32(t, what):
    if t.dtype != torch.float32:
        raise TypeError(f'{what} must be float32, got {t.dtype}')
    return t


def _is_nhwc(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=_CL)


def _dense(t, nhwc):
    """offset / mask tensors travel by strides; they only need to be non-overlapping and dense so
    that `empty_like` gives a gradient buffer with identical strides."""
    if t.is_contiguous() or t.is_contiguous(memory_format=_CL):
        return t
    return t.contiguous(memory_format=_CL) if nhwc else t.contiguous()


def _pixel_pitch(t):
    """Floats from one pixel to the next when `t` (B, C, H, W) is a channel slice of a dense channels-last tensor (or a
    dense one itself), else None."""
    if t.dim() != 4 or t.stride(1) != 1:
        return None
    B, C, H, W = t.shape
    p = t.stride(3)
    if p < C or p % 4 or t.stride(2) != W * p or (B > 1 and t.stride(0) != H * W * p):
        return None
    return p


def _strides(t):
    s = t.stride()
    return _lib.Strides4(s[0], s[1], s[2], s[3])


class HipBackend:
    name = 'hip'

    # ------------------------------------------------------------------ group norm (+ReLU), channels-last
    @staticmethod
    def group_norm_supported(C, G):
        return C % 4 == 0 and C % G == 0 and ((C // G) % 4 == 0 or 4 % (C // G) == 0) and C // 4 <= 256 and 256 % (C // 4) == 0 and G <= 256

    @staticmethod
    def _gn_levels(xs, ys=None, dys=None, dxs=None):
        n = len(xs)
        levels = (_lib.GnLevel * n)()
        for i, x in enumerate(xs):
            B, C, H, W = x.shape
            L = levels[i]
            L.x = _ptr(x)
            L.y = _ptr(ys[i]) if ys is not None else None
            L.dy = _ptr(dys[i]) if dys is not None else None
            L.dx = _ptr(dxs[i]) if dxs is not None else None
            L.B, L.HW = B, H * W
        return levels

    def group_norm_forward(self, xs, gamma, beta, groups, eps, relu, cat_px=False):
        """xs: channels-last (B, C, H, W) tensors sharing gamma/beta.  Returns (ys, mean_rstd).
        cat_px: the outputs are written as the levels of ONE (B, N_all, C) tensor (pixel rows of all levels back to back:
        LSHead._cat_px's layout); ys is then that tensor, viewed as (B, C, N_all, 1)."""
        lib = _lib.load()
        n, C = len(xs), xs[0].shape[1]
        if cat_px:
            B = xs[0].shape[0]
            assert all(x.shape[0] == B for x in xs)
            hw = [x.shape[2] * x.shape[3] for x in xs]
            flat = torch.empty((B, sum(hw), C), device=xs[0].device, dtype=torch.float32)
            ys, o = [], 0
            for k in hw:
                ys.append(flat[:, o:o + k])      # (B, k, C), batch stride N_all * C
                o += k
            levels = self._gn_levels(xs, ys=ys)
            for L in levels:
                L.y_batch_stride = flat.stride(0)
        else:
            ys = [torch.empty_like(x, memory_format=_CL) for x in xs]
            levels = self._gn_levels(xs, ys=ys)
        images = sum(x.shape[0] for x in xs)
        mean_rstd = torch.empty(images, groups, 2, device=xs[0].device, dtype=torch.float32)
        ws = torch.empty(lib.lsn_group_norm_workspace_bytes(n, levels, C, groups), device=xs[0].device,
                         dtype=torch.uint8)
        _lib.check(lib.lsn_group_norm_forward(n, levels, C, groups, _ptr(gamma), _ptr(beta), ctypes.c_float(eps),
                                              1 if relu else 0, _ptr(mean_rstd), _ptr(ws), _stream()))
        if cat_px:
            return flat.unsqueeze(2).permute(0, 3, 1, 2), mean_rstd      # (B, C, N_all, 1), channels-last memory
        return ys, mean_rstd

    supports_grad_sinks = True

    def group_norm_backward(self, xs, dys, gamma, beta, groups, relu, mean_rstd, need_params, sinks=None):
        """Returns (dxs, dgamma, dbeta); dgamma/dbeta are None unless need_params.  `sinks` = (dgamma, dbeta) buffers
        to ADD the parameter gradients to (ops/grad_sink.py)."""
        lib = _lib.load()
        n, C = len(xs), xs[0].shape[1]
        dxs = [torch.empty_like(x, memory_format=_CL) for x in xs]
        if torch.is_tensor(dys):     # the gradient of a cat_px output: (B, C, N_all, 1); its levels are read in place
            B = dys.shape[0]
            flat = dys.permute(0, 2, 3, 1).reshape(B, -1, C)
            if not flat.is_contiguous():
                flat = flat.contiguous()
            dyl, o = [], 0
            for x in xs:
                k = x.shape[2] * x.shape[3]
                dyl.append(flat[:, o:o + k])
                o += k
            levels = self._gn_levels(xs, dys=dyl, dxs=dxs)
            for L in levels:
                L.dy_batch_stride = flat.stride(0)
        else:
            dys = [d.contiguous(memory_format=_CL) for d in dys]
            levels = self._gn_levels(xs, dys=dys, dxs=dxs)
        if sinks:
            dg, db = sinks
        else:
            dg = torch.empty_like(gamma) if need_params else None
            db = torch.empty_like(beta) if need_params else None
        ws = torch.empty(lib.lsn_group_norm_workspace_bytes(n, levels, C, groups), device=xs[0].device,
                         dtype=torch.uint8)
        _lib.check(lib.lsn_group_norm_backward(n, levels, C, groups, _ptr(gamma), _ptr(beta), 1 if relu else 0,
                                               _ptr(mean_rstd), _ptr(dg), _ptr(db), _ptr(ws), 1 if sinks else 0,
                                               _stream()))
        return dxs, dg, db

    # ------------------------------------------------------------------ deformable conv family
    @staticmethod
    def _prep(inputs, offsets, masks, weight):
        nhwc = all(_is_nhwc(x) for x in inputs)
        if nhwc:
            xs = list(inputs)
            w = weight.contiguous(memory_format=_CL)
        else:
            xs = [x.contiguous() for x in inputs]
            w = weight.contiguous()
        offs = [_dense(_f32(o, 'offset'), nhwc) for o in offsets]
        msks = [None if m is None else _dense(_f32(m, 'mask'), nhwc) for m in masks]
        for x in xs:
            _f32(x, 'input')
        _f32(w, 'weight')
        return nhwc, xs, offs, msks, w

    @staticmethod
    def _shape(w, cfg):
        Co, Cg, kh, kw = w.shape
        return _lib.DcnShape(0, Cg * cfg['groups'], 0, 0, Co, 0, 0, kh, kw, cfg['stride'], cfg['pad'],
                             cfg['dil'], cfg['groups'], cfg['dg'], 1.0, 1.0, 1 if cfg.get('fused_om') else 0)

    @staticmethod
    def _mask_view_ptr(om, cfg, kk):
        """Fused DCNv2 pack: `om` (B, 3*dg*K, H, W) holds offsets in its first 2*dg*K channels and the
        mask logits in the rest; the mask pointer is the same buffer advanced by 2*dg*K channels."""
        return ctypes.c_void_p(om.data_ptr() + 2 * cfg['dg'] * kk * om.stride(1) * om.element_size())

    @staticmethod
    def _cached_image(w, weight, cfg, kind, nhwc):
        """The fragment image of a deformable layer's weight from the per-optimizer-step cache of the dense convolutions
        (ops/conv.py weight_image; kind 0: forward, 2: the backward GEMM) -- only for a channels-last PARAMETER handed over as it is
        (a temporary's image would be built and dropped per call: the in-call build is cheaper), groups = 1.  None: build per call.
        The library has the last word (lsn_dcn_prepared_ok)."""
        if not (CACHE_DCN_IMAGES and nhwc and w is weight and cfg['groups'] == 1 and isinstance(weight, torch.nn.Parameter)):
            return None
        from .conv import weight_image
        try:
            return weight_image(w, kind, 1, cfg['pad'], cfg['dil'])
        except NotImplementedError:
            return None

    def dcn_forward(self, inputs, offsets, masks, weight, bias, cfg, out_hw):
        """inputs/offsets/masks: per-level lists (mask entries may be None); cfg: stride, pad, dil,
        groups, dg, scales[(sh, sw)]; out_hw: per-level (Ho, Wo).  Returns the per-level outputs."""
        lib = _lib.load()
        nhwc, xs, offs, msks, w = self._prep(inputs, offsets, masks, weight)
        n = len(xs)
        shape = self._shape(w, cfg)
        levels = (_lib.DcnLevel * n)()
        outs = []
        # cfg['concat'] = g: the outputs of g consecutive levels are wanted side by side in ONE (B, g Co, Ho, Wo) tensor (LSHead:
        # the three maps a level gathers, lsnet_head.py:640-647).  The kernels write them there (lsn_dcn_shape.out_pitch)
        # when they can; otherwise the levels get their own buffers and ATen concatenates.
        g = int(cfg.get('concat') or 0)
        Co = w.shape[0]
        wide = None
        if g > 1:
            assert n % g == 0 and all(out_hw[i] == out_hw[i - i % g] and xs[i].shape[0] == xs[i - i % g].shape[0] for i in range(n))
            if nhwc and (g * Co) % 4 == 0:
                wide = [torch.empty((xs[j].shape[0], g * Co) + tuple(out_hw[j]), device=xs[j].device, dtype=torch.float32,
                                    memory_format=_CL) for j in range(0, n, g)]
        for i in range(n):
            B, C, H, W = xs[i].shape
            Ho, Wo = out_hw[i]
            if wide is not None:
                out = wide[i // g].narrow(1, (i % g) * Co, Co)
            else:
                out = torch.empty((B, Co, Ho, Wo), device=xs[i].device, dtype=torch.float32,
                                  memory_format=_CL if nhwc else torch.contiguous_format)
            outs.append(out)
            L = levels[i]
            L.input, L.offset, L.mask, L.output = _ptr(xs[i]), _ptr(offs[i]), _ptr(msks[i]), _ptr(out)
            L.off_st = _strides(offs[i])
            if cfg.get('fused_om'):
                L.mask, L.mask_st = self._mask_view_ptr(offs[i], cfg, w.shape[2] * w.shape[3]), _strides(offs[i])
            elif msks[i] is not None:
                L.mask_st = _strides(msks[i])
            L.B, L.H, L.W, L.Ho, L.Wo = B, H, W, Ho, Wo
            L.scale_h, L.scale_w = cfg['scales'][i]
        b = None if bias is None else _f32(bias.contiguous(), 'bias')
        if _lib.split_math():
            img = self._cached_image(w, weight, cfg, 0, nhwc)
            if img is not None:
                shape.workspace, shape.weights_prepared = img.data_ptr(), 1
                if not lib.lsn_dcn_prepared_ok(ctypes.byref(shape), n, levels, 0):
                    img, shape.weights_prepared = None, 0
            if img is None:
                ws = torch.empty(2 * w.numel(), device=w.device, dtype=torch.float32)   # pre-split weight planes
                shape.workspace = ws.data_ptr()
        if wide is not None:
            shape.out_pitch = g * Co
            if not lib.lsn_dcn_pitched_ok(ctypes.byref(shape), n, levels, 0):   # (exact-fp32 mode, odd channel counts)
                shape.out_pitch = 0
                wide = None
                outs = [torch.empty(tuple(o.shape), device=o.device, dtype=torch.float32, memory_format=_CL) for o in outs]
                for i in range(n):
                    levels[i].output = _ptr(outs[i])
        _lib.check(lib.lsn_dcn_forward(ctypes.byref(shape), n, levels, _ptr(w), _ptr(b), 1 if nhwc else 0,
                                       _stream()))
        if g > 1:
            return wide if wide is not None else [torch.cat(outs[j:j + g], dim=1) for j in range(0, n, g)]
        return outs

    def dcn_backward(self, inputs, offsets, masks, weight, grad_outs, cfg, need):
        """need: dict(input=[bool]*n, offset=[bool]*n, mask=[bool]*n, weight=bool, bias=bool).
        Returns (grad_inputs, grad_offsets, grad_masks, grad_weight, grad_bias)."""
        nhwc, xs, offs, msks, w = self._prep(inputs, offsets, masks, weight)
        n = len(xs)
        # channel slices of wider channels-last tensors (the gradient of a concatenation) are read where they lie when the
        # kernels can (lsn_dcn_shape.out_pitch): _dcn_backward_call falls back to dense copies otherwise
        pitches = {_pixel_pitch(g) for g in grad_outs} if nhwc else {None}
        if len(pitches) == 1 and None not in pitches and pitches != {w.shape[0]}:
            gos = list(grad_outs)
        else:
            gos = [(g.contiguous(memory_format=_CL) if nhwc else g.contiguous()) for g in grad_outs]
        gxs, goffs, gmsks = [], [], []
        shared = {}   # levels sampling ONE source map (pyramid op) accumulate into one grad_input buffer
        bufs = []     # per level: the grad_input buffer the kernels write (shared ones appear several times)
        for i in range(n):
            gx = gx_ret = None
            if need['input'][i]:
                key = (xs[i].data_ptr(), tuple(xs[i].shape)) if nhwc else i
                if key in shared:
                    gx = shared[key]            # the first level of the group returns the buffer, the others None
                else:
                    gx = gx_ret = shared[key] = torch.empty_like(xs[i])
            want_om = need['offset'][i] or (msks[i] is not None and need['mask'][i])
            goffs.append(torch.empty_like(offs[i]) if want_om else None)
            gmsks.append(torch.empty_like(msks[i]) if (want_om and msks[i] is not None) else None)
            gxs.append(gx_ret)
            bufs.append(gx)
        sinks = need.get('sinks') if (nhwc and w is weight) else None
        if sinks is not None:     # accumulate into the caller's gradient arena (ops/grad_sink.py)
            gw, gb = sinks
            acc = 1
            need['sunk'] = True
        else:
            gw = torch.empty_like(w) if (need['weight'] or need['bias']) else None
            gb = torch.empty(w.shape[0], device=w.device, dtype=torch.float32) if need['bias'] else None
            acc = 0
        # (One call per image -- the column gradients of ONE image of a tower launch, 207 MB, fit the 256 MB Infinity Cache,
        # those of the batch, 413 MB, do not -- was measured SLOWER: deformable backward-data 7.6 vs 6.8 ms per step,
        # profiles/r4_per_image.txt: half-size launches lose more to their tails than the cache returns.)
        self._dcn_backward_call(xs, offs, msks, w, gos, bufs, goffs, gmsks, gw, gb, acc, cfg, nhwc, weight)
        return gxs, goffs, gmsks, gw, gb

    def _dcn_backward_call(self, xs, offs, msks, w, gos, gx_bufs, goffs, gmsks, gw, gb, accumulate, cfg, nhwc, weight=None):
        """One lsn_dcn_backward call over the given (level) tensors; every output buffer is the caller's."""
        lib = _lib.load()
        n = len(xs)
        shape = self._shape(w, cfg)
        levels = (_lib.DcnLevel * n)()
        for i in range(n):
            B, C, H, W = xs[i].shape
            go, gx, goff, gmsk = gos[i], gx_bufs[i], goffs[i], gmsks[i]
            Ho, Wo = go.shape[2], go.shape[3]
            L = levels[i]
            L.input, L.offset, L.mask = _ptr(xs[i]), _ptr(offs[i]), _ptr(msks[i])
            L.grad_output, L.grad_input, L.grad_offset, L.grad_mask = _ptr(go), _ptr(gx), _ptr(goff), _ptr(gmsk)
            L.off_st = _strides(offs[i])
            if cfg.get('fused_om'):
                kk = w.shape[2] * w.shape[3]
                L.mask, L.mask_st = self._mask_view_ptr(offs[i], cfg, kk), _strides(offs[i])
                if goff is not None:   # one gradient tensor for offsets + mask logits
                    L.grad_mask = self._mask_view_ptr(goff, cfg, kk)
            elif msks[i] is not None:
                L.mask_st = _strides(msks[i])
            L.B, L.H, L.W, L.Ho, L.Wo = B, H, W, Ho, Wo
            L.scale_h, L.scale_w = cfg['scales'][i]
        ws = gws = None
        img = self._cached_image(w, weight, cfg, 2, nhwc) if _lib.split_math() else None
        if img is not None:
            shape.workspace = img.data_ptr()       # (weights_prepared is decided below, when the gather workspace is known)
        elif cfg['groups'] == 1 and _lib.split_math():
            ws = torch.empty(2 * w.numel(), device=w.device, dtype=torch.float32)   # split + transposed weight planes
            shape.workspace = ws.data_ptr()
        # column-gradient buffer + anchor lists of the atomic-free grad_input path, in every math mode (round 6: the exact mode's
        # column gradients are fmaf chains in front of the same gather; 0 bytes: a shape only the scatter kernels serve)
        nbytes = int(lib.lsn_dcn_backward_workspace_bytes(ctypes.byref(shape), n, levels))
        if nbytes > 0:
            gws = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
            shape.gather_workspace, shape.gather_workspace_bytes = gws.data_ptr(), nbytes
        if img is not None:
            shape.weights_prepared = 1
            if not lib.lsn_dcn_prepared_ok(ctypes.byref(shape), n, levels, 1):   # (the cached image must not be written: own scratch)
                shape.weights_prepared = 0
                ws = torch.empty(2 * w.numel(), device=w.device, dtype=torch.float32)
                shape.workspace = ws.data_ptr()
        shape.accumulate_param_grads = 1 if accumulate else 0
        pitch = _pixel_pitch(gos[0]) if nhwc else None
        if pitch is not None and pitch != w.shape[0]:
            shape.out_pitch = pitch
            if not lib.lsn_dcn_pitched_ok(ctypes.byref(shape), n, levels, 1):
                shape.out_pitch = 0
                gos = [g.contiguous(memory_format=_CL) for g in gos]
                for i in range(n):
                    levels[i].grad_output = _ptr(gos[i])
        _lib.check(lib.lsn_dcn_backward(ctypes.byref(shape), n, levels, _ptr(w), _ptr(gw), _ptr(gb),
                                        1 if nhwc else 0, _stream()))

    # ------------------------------------------------------------------ sigmoid focal loss
    def focal_forward(self, logits, targets, gamma, alpha):
        lib = _lib.load()
        logits = _f32(logits, 'logits').contiguous()
        targets = targets.contiguous()
        if targets.dtype != torch.int64:
            raise TypeError('targets must be int64')
        out = torch.empty_like(logits)
        N, C = logits.shape
        _lib.check(lib.lsn_sigmoid_focal_loss_forward(_ptr(logits), _ptr(targets), _ptr(out), N, C,
                                                      ctypes.c_float(gamma), ctypes.c_float(alpha), _stream()))
        return out

    def focal_backward(self, logits, targets, d_losses, gamma, alpha):
        lib = _lib.load()
        logits = logits.contiguous()
        d_losses = d_losses.contiguous()
        out = torch.empty_like(logits)
        N, C = logits.shape
        _lib.check(lib.lsn_sigmoid_focal_loss_backward(_ptr(logits), _ptr(targets.contiguous()), _ptr(d_losses),
                                                       _ptr(out), N, C, ctypes.c_float(gamma),
                                                       ctypes.c_float(alpha), _stream()))
        return out

    def focal_sum(self, logits, targets, weight, gamma, alpha):
        lib = _lib.load()
        logits = _f32(logits, 'logits').contiguous()
        out = torch.empty((), device=logits.device, dtype=torch.float32)
        N, C = logits.shape
        w = None if weight is None else _f32(weight, 'weight').contiguous()
        _lib.check(lib.lsn_sigmoid_focal_loss_sum(_ptr(logits), _ptr(targets.contiguous()), _ptr(w), _ptr(out), N,
                                                  C, ctypes.c_float(gamma), ctypes.c_float(alpha), _stream()))
        return out

    def focal_backward_weighted(self, logits, targets, weight, scale, gamma, alpha):
        lib = _lib.load()
        logits = logits.contiguous()
        out = torch.empty_like(logits)
        N, C = logits.shape
        w = None if weight is None else weight.contiguous()
        scale = scale.to(torch.float32).contiguous()
        _lib.check(lib.lsn_sigmoid_focal_loss_backward_weighted(
            _ptr(logits), _ptr(targets.contiguous()), _ptr(w), _ptr(scale), _ptr(out), N, C,
            ctypes.c_float(gamma), ctypes.c_float(alpha), _stream()))
        return out

    # ------------------------------------------------------------------ per-level loss sums (LSHead's concatenated rows)
    @staticmethod
    def _level_starts(num_level):
        starts = [0]
        for n in num_level:
            starts.append(starts[-1] + int(n))
        return (ctypes.c_int * len(starts))(*starts), starts[-1]

    def focal_level_sums(self, logits, targets, weight, B, num_level, gamma, alpha):
        """(L,) per-level sums of weight * focal loss over the (B * N_all, C) rows, levels back to back per image."""
        lib = _lib.load()
        assert logits.is_contiguous() and targets.is_contiguous() and (weight is None or weight.is_contiguous())
        _f32(logits, 'logits')
        if targets.dtype != torch.int64:
            raise TypeError('targets must be int64')
        starts, nall = self._level_starts(num_level)
        N, C = logits.shape
        assert N == B * nall and targets.numel() == N
        out = torch.empty(len(num_level), device=logits.device, dtype=torch.float32)
        _lib.check(lib.lsn_sigmoid_focal_loss_level_sums(_ptr(logits), _ptr(targets), _ptr(weight), _ptr(out), B, nall, C,
                                                         len(num_level), starts, ctypes.c_float(gamma), ctypes.c_float(alpha),
                                                         _stream()))
        return out

    def focal_backward_levels(self, logits, targets, weight, scales, B, num_level, gamma, alpha):
        lib = _lib.load()
        starts, nall = self._level_starts(num_level)
        N, C = logits.shape
        out = torch.empty_like(logits)
        scales = scales.to(torch.float32).contiguous()
        assert scales.numel() == len(num_level)
        _lib.check(lib.lsn_sigmoid_focal_loss_backward_levels(_ptr(logits), _ptr(targets), _ptr(weight), _ptr(scales), _ptr(out),
                                                              B, nall, C, len(num_level), starts, ctypes.c_float(gamma),
                                                              ctypes.c_float(alpha), _stream()))
        return out

    def level_sums(self, rows, B, num_level):
        lib = _lib.load()
        starts, nall = self._level_starts(num_level)
        assert rows.is_contiguous() and rows.numel() == B * nall and rows.dtype == torch.float32
        out = torch.empty(len(num_level), device=rows.device, dtype=torch.float32)
        _lib.check(lib.lsn_level_sums(_ptr(rows), _ptr(out), B, nall, len(num_level), starts, _stream()))
        return out

    def level_expand(self, g, B, num_level):
        lib = _lib.load()
        starts, nall = self._level_starts(num_level)
        g = g.to(torch.float32).contiguous()
        out = torch.empty(B * nall, device=g.device, dtype=torch.float32)
        _lib.check(lib.lsn_level_expand(_ptr(g), _ptr(out), B, nall, len(num_level), starts, _stream()))
        return out

    # ------------------------------------------------------------------ NMS
    def nms(self, dets, iou_thr):
        """dets (n,5) float32 on the device -> keep indices (int64), descending score."""
        lib = _lib.load()
        dets = _f32(dets, 'dets').contiguous()
        n = dets.shape[0]
        if n == 0:
            return dets.new_zeros(0, dtype=torch.long)
        order = dets[:, 4].sort(0, descending=True)[1].contiguous()  # nms_kernel.cu:81-83
        keep = torch.empty(n, dtype=torch.int64, device=dets.device)
        num = torch.zeros(1, dtype=torch.int64, device=dets.device)
        ws = torch.empty(int(lib.lsn_nms_workspace_bytes(n)), dtype=torch.uint8, device=dets.device)
        _lib.check(lib.lsn_nms(_ptr(dets), _ptr(order), n, ctypes.c_float(iou_thr), _ptr(keep), _ptr(num),
                               _ptr(ws), _stream()))
        return keep[:int(num.item())]

    # ------------------------------------------------------------------ k nearest per column (assigners)
    def topk_columns(self, x, k, seg_start, seg_len, largest=False):
        """x (P, G) float32 on the device -> (values, indices), both (nseg * k, G): per row segment the k smallest
        (largest) entries of every column in that order -- torch.topk(x[s:s + n], k, dim=0) of every segment, one launch."""
        lib = _lib.load()
        x = _f32(x, 'x')
        if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
            x = x.contiguous()
        P, G = x.shape
        nseg = len(seg_start)
        vals = torch.empty((nseg * k, G), dtype=torch.float32, device=x.device)
        idx = torch.empty((nseg * k, G), dtype=torch.int64, device=x.device)
        starts = (ctypes.c_int * nseg)(*[int(v) for v in seg_start])
        lens = (ctypes.c_int * nseg)(*[int(v) for v in seg_len])
        _lib.check(lib.lsn_topk_columns(_ptr(x), P, G, x.stride(0), nseg, starts, lens, int(k), 1 if largest else 0,
                                        _ptr(vals), _ptr(idx), _stream()))
        return vals, idx

    # ------------------------------------------------------------------ LSHead's cumulative offset rescaling
    @staticmethod
    def offset_chain_ok(off):
        """(B, C, H, W) fp32 with channels innermost and pixels back to back inside an image (any image pitch)."""
        if not (off.is_cuda and off.dtype == torch.float32 and off.dim() == 4 and off.shape[1] % 2 == 0):
            return False
        B, C, H, W = off.shape
        return off.stride(1) == 1 and off.stride(3) == C and (H == 1 or off.stride(2) == W * C) and \
            (B == 1 or off.stride(0) >= H * W * C)

    def offset_chain(self, offs, mults):
        """offs: per-level offset fields; mults[l] = ((sh, sw) x 3) -> per level the three rescaled fields."""
        lib = _lib.load()
        n, C = len(offs), offs[0].shape[1]
        lv = (_lib.OffsetChainLevel * n)()
        res = []
        for l, off in enumerate(offs):
            B, _, H, W = off.shape
            L = lv[l]
            L.images, L.per_image = B, H * W * C
            L.off, L.off_image_pitch = _ptr(off), (off.stride(0) if B > 1 else H * W * C)
            outs = [torch.empty((B, C, H, W), device=off.device, dtype=torch.float32, memory_format=_CL) for _ in range(3)]
            for k in range(3):
                L.mh[k], L.mw[k] = mults[l][k]
                L.out[k] = outs[k].data_ptr()
            res.append(outs)
        _lib.check(lib.lsn_offset_chain_forward(n, lv, C, _stream()))
        return res

    def offset_chain_backward(self, shapes, device, mults, grads):
        """grads[l]: the gradients of level l's three fields, then (optionally) of a second consumer's aliases of them --
        3 or 6 entries, any of them None -> per level the offset field's gradient."""
        lib = _lib.load()
        n, C = len(shapes), shapes[0][1]
        lv = (_lib.OffsetChainLevel * n)()
        keep, res = [], []
        for l, (B, _, H, W) in enumerate(shapes):
            L = lv[l]
            L.images, L.per_image = B, H * W * C
            for k in range(3):
                L.mh[k], L.mw[k] = mults[l][k]
            for k, g in enumerate(grads[l]):
                if g is not None:
                    g = _f32(g, 'grad').contiguous(memory_format=_CL)
                    keep.append(g)
                    L.gout[k] = g.data_ptr()
            goff = torch.empty((B, C, H, W), device=device, dtype=torch.float32, memory_format=_CL)
            L.goff = goff.data_ptr()
            res.append(goff)
        _lib.check(lib.lsn_offset_chain_backward(n, lv, C, _stream()))
        return res

    def selftest_mfma(self, A, B, variant):
        lib = _lib.load()
        A, B = A.contiguous(), B.contiguous()
        M, K = A.shape
        N = B.shape[1]
        D = torch.empty(M, N, device=A.device, dtype=torch.float32)
        _lib.check(lib.lsn_selftest_mfma(_ptr(A), _ptr(B), _ptr(D), M, N, K, variant, _stream()))
        return D


_lib.load()  # fail loudly at registration time when the library is missing
register_backend('cuda', HipBackend())
