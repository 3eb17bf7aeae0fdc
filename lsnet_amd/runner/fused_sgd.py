"""Gradient clipping + SGD step in three launches (include/lsnet_hip.h: lsn_clip_sgd_step) for the runner's OptimizerHook.

The reference's iteration ends with `clip_grad_norm_(params, max_norm=35, norm_type=2)` and `optimizer.step()`
(mmcv/runner/hooks/optimizer.py:8-28): in ATen a norm per tensor, the norm of those, a clamp, a multiplication per tensor and
the multi-tensor SGD kernels -- a dozen launches over ~210 tensors, 0.36 ms of the LSNet R-50 step.  When the gradients
live where they will still live next step (the all-reduce buckets of parallel/reducer.py: `p.grad` is a stable view) the
hook hands the whole thing to the library: per element the same operations in the same rounding order, the clip
coefficient formed on the device.  Anything else -- no buckets, Nesterov / dampening / maximize, another optimizer, a norm
type other than 2, CPU tensors -- keeps the torch path."""
import ctypes

import torch

from .. import _lib
from ..ops import conv as _conv


def _dense_like(a, b):
    return a.dtype == torch.float32 and a.is_cuda and a.stride() == b.stride() and a.shape == b.shape


def _signature(optimizer):
    """What a declined plan was declined on: the gradient and momentum-buffer storage of every parameter.  A plan that was
    declined (gradients not there yet, not yet views of the buckets) is rebuilt when this changes, not cached for ever."""
    sig = []
    for grp in optimizer.param_groups:
        for p in grp['params']:
            g = p.grad
            buf = optimizer.state.get(p, {}).get('momentum_buffer') if p in optimizer.state else None
            sig.append((0 if g is None else g.data_ptr(), 0 if buf is None else buf.data_ptr()))
    return tuple(sig)


class ClipSGD:
    """Built once per (optimizer, parameter set); `step()` re-validates the pointers it baked into the device table.
    The fused step replaces `clip_grad_norm_` + `optimizer.step()`: optimizer step hooks (register_step_pre_hook / _post_hook)
    and `_step_count` are NOT driven by it, and the norm is taken over the optimizer's parameters (the reference clips
    `model.parameters()`: the same set in every LSNet config)."""

    def __init__(self, optimizer, grad_clip):
        self.opt, self.ok = optimizer, False
        self.signature = _signature(optimizer)
        self.max_norm = 0.0
        if grad_clip is not None:
            if float(grad_clip.get('norm_type', 2)) != 2.0 or grad_clip.get('error_if_nonfinite') or grad_clip.get('foreach') is False:
                return
            self.max_norm = float(grad_clip['max_norm'])
            if not self.max_norm > 0:
                return
        if type(optimizer) is not torch.optim.SGD or len(optimizer.param_groups) > 8:
            return
        entries, chunk = [], 0
        for gi, grp in enumerate(optimizer.param_groups):
            if grp.get('nesterov') or grp.get('dampening', 0) != 0 or grp.get('maximize') or not grp.get('momentum', 0) > 0 \
                    or grp.get('differentiable'):
                return
            for p in grp['params']:
                if not p.requires_grad:
                    continue
                g = p.grad
                if g is None or not p.is_cuda or p.dtype != torch.float32 or not _dense_like(g, p) or p.numel() == 0:
                    return
                if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                    return
                buf = optimizer.state[p].get('momentum_buffer') if p in optimizer.state else None
                if buf is not None and not _dense_like(buf, p):
                    return
                if (p.data_ptr() | g.data_ptr() | (0 if buf is None else buf.data_ptr())) & 15:
                    return
                entries.append([p, g, buf, gi, chunk])
                chunk += (p.numel() + 4095) // 4096
        if not entries:
            return
        # every check has passed: only now touch the optimizer's state (a plan that went on to decline had already
        # written zero momentum buffers).  torch's first step clones the gradient: zeros give the same buffer (buf m + d = d)
        for e in entries:
            if e[2] is None:
                e[2] = optimizer.state[e[0]]['momentum_buffer'] = torch.zeros_like(e[0])
                if e[2].data_ptr() & 15:
                    return
        entries = [tuple(e) for e in entries]
        self.entries, self.chunks = entries, chunk
        dev = entries[0][0].device
        host = (_lib.SgdTensor * len(entries))()
        for h, (p, g, buf, gi, c0) in zip(host, entries):
            h.param, h.grad, h.momentum_buf, h.numel, h.first_chunk, h.group = p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), c0, gi
        raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8)
        self.table = raw.to(dev)                         # (uploaded once: the pointers are stable by construction)
        self.ptrs = [(p.data_ptr(), g.data_ptr(), buf.data_ptr()) for p, g, buf, _, _ in entries]
        self.ws = torch.empty(int(_lib.load().lsn_clip_sgd_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.stats = torch.zeros(2, dtype=torch.float32, device=dev)
        self.groups = (_lib.SgdGroup * len(optimizer.param_groups))()
        self.ok = True

    def still_valid(self):
        for (p, g, buf, _, _), (pp, gp, bp) in zip(self.entries, self.ptrs):
            if p.grad is not g or p.data_ptr() != pp or g.data_ptr() != gp or self.opt.state[p].get('momentum_buffer') is not buf:
                return False
        return True

    def step(self):
        """-> the total gradient norm (device scalar) when clipping, else None."""
        lib = _lib.load()
        for q, grp in zip(self.groups, self.opt.param_groups):
            q.lr, q.momentum, q.weight_decay = float(grp['lr']), float(grp['momentum']), float(grp['weight_decay'])
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.lsn_clip_sgd_step(len(self.entries), ctypes.c_void_p(self.table.data_ptr()), self.chunks, len(self.groups),
                                         self.groups, ctypes.c_float(self.max_norm), ctypes.c_void_p(self.ws.data_ptr()),
                                         ctypes.c_void_p(self.stats.data_ptr()), stream))
        _conv.parameters_updated()      # the cached weight images of the convolutions are stale now
        return self.stats[0] if self.max_norm > 0 else None
