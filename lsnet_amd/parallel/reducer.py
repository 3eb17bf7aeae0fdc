"""Data-parallel gradient averaging over RCCL (one process per GPU), overlapped with backward.

Takes the place of the reference's `MMDistributedDataParallel` = torch DDP over NCCL
(mmcv/parallel/distributed.py:10-53, mmdet/apis/train.py:74-78).  Design for xGMI:
  * gradients are packed into few large flat buckets in REVERSE registration order (head -> neck ->
    backbone, the order backward produces them); xGMI is point-to-point (7 links x ~153 GB/s), so a
    ring all-reduce is bound by one link: large buckets amortise latency, and 154 MB of fp32 grads
    (R-50) takes ~2 ms per step, far below the compute time -- it only has to start early;
  * `p.grad` is a VIEW into its bucket, so backward writes the communication buffers directly (no copies);
  * a bucket's all-reduce is launched (async, on RCCL's own stream) the moment its last gradient
    has been accumulated, from `register_post_accumulate_grad_hook`, so it overlaps the rest of
    backward; `finish()` waits and averages in place before clip-grad / SGD.
Works with any torch.distributed backend ('nccl' = RCCL on ROCm; 'gloo' in the CPU tests)."""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from ..ops import grad_sink


def init_dist(backend='nccl', **kwargs):
    """`init_dist('pytorch')` of the reference (mmcv/runner/dist_utils.py:26-31): env:// rendezvous,
    one process per GPU."""
    rank = int(os.environ.get('RANK', 0))
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank % max(torch.cuda.device_count(), 1))))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kwargs)
    return dist.get_rank(), dist.get_world_size()


# arena size (MB of partial tiles) at which the library runs its queued weight-gradient reduces on its own; 0: no deferral
WGRAD_DEFER_MB = int(os.environ.get('LSNET_WGRAD_DEFER_MB', '4096'))

class BucketedGradReducer:
    """Gradient-VIEW buckets: every `p.grad` is a view (in the parameter's own memory layout) into one of a few flat
    tensors, so backward accumulates straight into the communication buffers -- no gradient -> bucket -> gradient
    copies, one fill per bucket to zero them, one all-reduce per bucket.  The views are also registered as the
    parameters' gradient SINKS (ops/grad_sink.py): the package's weight-gradient kernels add into them directly, so a
    step has no per-parameter gradient allocation, memset or `p.grad.add_()` either.  With world size 1 the class is
    just that gradient arena (no communication).

    Buckets are all-reduced strictly in index order (the order backward completes them: head -> neck -> backbone):
    a bucket is launched during backward only when every earlier bucket has been launched, everything else in
    `finish()`.  All ranks therefore issue the same collectives in the same order whatever subset of parameters
    received a gradient.  A parameter is "complete" when it has received as many gradient contributions (autograd's
    AccumulateGrad, or a kernel's direct accumulation into the sink) as in the previous step -- the training graph is
    static; a contribution that arrives after its bucket went out raises instead of being lost.  Parameters without a
    gradient in a step keep zeros in their slot (with torch DDP they would keep `grad=None`; every trainable LSNet
    parameter gets a gradient in every step, frozen ones have requires_grad=False and are not bucketed)."""

    def __init__(self, params, bucket_mb=25.0, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # collectives are skipped on a single rank -- unless asked for: LSNET_FORCE_COLLECTIVES=1 sends the buckets
        # through a one-rank group all the same, which is how a 1-GPU box executes the RCCL side of this class
        # (async work objects, stream ordering against the kernels that fill the buckets, the wait in finish())
        self.collective = self.world > 1 or (dist.is_initialized() and os.environ.get('LSNET_FORCE_COLLECTIVES') == '1')
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []          # dict(flat, params, views, pending, work)
        cap = int(bucket_mb * 1024 * 1024 / 4)
        cur, size = [], 0
        for p in reversed(self.params):
            if cur and size + self._slot(p.numel()) > cap:
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += self._slot(p.numel())
        if cur:
            self._close(cur)
        self._one_arena()
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b['params']):
                self._where[p] = (bi, pi)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._deferring = False
        self._expected = None      # contributions per parameter and step, learnt from the previous step
        self._events = {}
        self._zeroed = False
        self.reset()

    def _one_arena(self):
        """The buckets as slices of ONE allocation (when they agree in dtype and device): zero_grad is one fill instead of one
        per bucket.  Every bucket starts on a 256-byte boundary."""
        if len(self.buckets) < 2 or len({(b['flat'].dtype, b['flat'].device) for b in self.buckets}) != 1:
            self.arena = None
            return
        starts, o = [], 0
        for b in self.buckets:
            starts.append(o)
            o += (b['flat'].numel() + 63) // 64 * 64
        self.arena = torch.zeros(o, dtype=self.buckets[0]['flat'].dtype, device=self.buckets[0]['flat'].device)
        for b, st in zip(self.buckets, starts):
            flat = self.arena[st:st + b['flat'].numel()]
            # (a view's storage offset was its offset inside the bucket's own allocation)
            b['views'] = [flat[v.storage_offset():].as_strided(v.shape, v.stride()) for v in b['views']]
            b['flat'] = flat

    @staticmethod
    def _slot(n):
        """Floats a parameter occupies in its bucket: rounded up to 16 bytes, so that every gradient sink starts on a
        16-byte boundary (the weight-gradient reduce kernels load and store float4 on it) whatever the odd-sized
        parameters (27-channel biases, ...) in front of it.  The pad floats stay zero and travel with the bucket."""
        return (n + 3) // 4 * 4

    def _close(self, plist):
        n = sum(self._slot(p.numel()) for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        views, o = [], 0
        for p in plist:
            seg = flat[o:o + p.numel()]
            # same memory layout as the parameter (channels-last conv weights): autograd then adds in place
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            views.append(seg.as_strided(p.shape, p.stride()) if dense else seg.view_as(p))
            o += self._slot(p.numel())
        self.buckets.append(dict(flat=flat, params=list(plist), views=views, pending=0, work=None))

    def reset(self):
        self._next = 0            # first bucket not yet launched
        for b in self.buckets:
            b['pending'] = len(b['params'])
            b['work'] = None
        self._events = {}

    @staticmethod
    def _is_view(g, v):
        return g is v or (g is not None and g.data_ptr() == v.data_ptr() and g.stride() == v.stride())

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): one fill per bucket; p.grad (re)pointed at its bucket view, which is also
        the parameter's gradient sink."""
        if getattr(self, 'arena', None) is not None:
            self.arena.zero_()
        for b in self.buckets:
            if getattr(self, 'arena', None) is None:
                b['flat'].zero_()
            for p, v in zip(b['params'], b['views']):
                if p.grad is not v:
                    p.grad = v
                if getattr(p, '_lsn_sink', None) is not v:
                    grad_sink.register(p, v, self._on_sink)
        self._zeroed = True
        from ..ops import resblock
        resblock.reset_pending()
        self._defer(True)

    def _defer(self, on):
        """The gradients of this step live in the buckets and nobody reads them before finish() (or a bucket's all-reduce):
        the library may keep the partial tiles of its weight-gradient kernels and reduce many gradients per launch
        (include/lsnet_hip.h lsn_wgrad_defer; 48 reduce launches of 15 - 19 us per step otherwise).  on = None: flush only."""
        if WGRAD_DEFER_MB <= 0 or not self.buckets or not self.buckets[0]['flat'].is_cuda:
            return
        from .. import _lib
        if on is None:
            if self._deferring:
                _lib.wgrad_flush()
        elif on:
            _lib.wgrad_defer(WGRAD_DEFER_MB)
            self._deferring = True
        elif self._deferring:
            _lib.wgrad_defer(0)
            self._deferring = False

    def _launch_ready(self):
        if not self.collective:
            return
        while self._next < len(self.buckets) and self.buckets[self._next]['pending'] == 0:
            b = self.buckets[self._next]
            from ..ops import resblock
            resblock.join_side(b['flat'].device if b['flat'].is_cuda else None)   # (weight gradients issued on the second stream)
            self._defer(None)     # queued weight-gradient reduces write into this bucket: run them before it leaves
            b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._next += 1

    def _event(self, p):
        """One gradient contribution to p has been accumulated into its bucket."""
        bi, _ = self._where[p]
        if bi < self._next:
            raise RuntimeError('BucketedGradReducer: a gradient arrived after its bucket was all-reduced -- the number of '
                               'gradient contributions per parameter changed between steps (or a second backward() ran '
                               'before finish()); dynamic graphs are not supported')
        c = self._events.get(p, 0) + 1
        self._events[p] = c
        if self._expected is not None and c == self._expected.get(p, -1):
            b = self.buckets[bi]
            b['pending'] -= 1
            if b['pending'] == 0:
                self._launch_ready()

    def _on_sink(self, p):
        self._event(p)

    def _on_grad(self, p):
        bi, pi = self._where[p]
        v = self.buckets[bi]['views'][pi]
        if not self._is_view(p.grad, v):   # the gradient was not preset (zero_grad() skipped): move it into the bucket
            if not self._zeroed and self._events.get(p, 0) == 0:
                v.copy_(p.grad)
            else:
                v.add_(p.grad)
            p.grad = v
        self._event(p)

    def finish(self):
        """Launch what backward could not (in order), wait, average."""
        from ..ops import resblock
        left = resblock.pending_wgrad_jobs()
        if left:   # (the engine callback of ops/resblock.py flushes at the end of every backward pass: this is a bug trap)
            raise RuntimeError(f'{left} queued weight-gradient job(s) of fused ResNet stages were never launched: their '
                               'parameters would be reduced without these contributions')
        if self.buckets and self.buckets[0]['flat'].is_cuda:
            resblock.join_side(self.buckets[0]['flat'].device)
        self._defer(False)   # every deferred reduce of the step runs now, ahead of whatever reads the buckets
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            for p, v in zip(b['params'], b['views']):
                if p.grad is None:
                    if not self._zeroed and self._events.get(p, 0) == 0:
                        v.zero_()     # no zero_grad() before this step: the slot still holds the previous gradient
                    p.grad = v
                elif not self._is_view(p.grad, v):
                    v.copy_(p.grad)
                    p.grad = v
            b['pending'] = 0
            if not self.collective:
                self._next += 1
            else:
                self._launch_ready()
        if self.collective:
            for b in self.buckets:
                b['work'].wait()
                if self.world > 1:
                    b['flat'].div_(self.world)
        self._expected = dict(self._events)
        self._zeroed = False
        self.reset()


class DataParallelModel(nn.Module):
    """`model.train_step(data, optimizer)` wrapper with the reference's interface
    (MMDistributedDataParallel.train_step): forwards to the wrapped detector; gradient averaging is
    started by hooks during backward and completed by `reduce_gradients()` (called by OptimizerHook
    right after `loss.backward()`)."""

    def __init__(self, module, bucket_mb=25.0, broadcast_params=True):
        super().__init__()
        self.module = module
        p0 = next(module.parameters(), None)
        if p0 is not None and p0.is_cuda:
            # the step's second stream (ops/streams.py) before the first collective: RCCL creates streams of its own there, and
            # which hardware queue a stream shares depends on the order of creation
            from ..ops.streams import side_stream, warm_library_streams
            side_stream(p0.device)
            warm_library_streams(p0.device)     # (the library's own second stream, likewise)
        if dist.is_initialized() and dist.get_world_size() > 1 and broadcast_params:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0)
            from ..ops.conv import invalidate_weight_images
            invalidate_weight_images()    # (written through .data: invisible to the image cache's version checks)
        self.reducer = BucketedGradReducer(module.parameters(), bucket_mb)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def train_step(self, *args, **kwargs):
        return self.module.train_step(*args, **kwargs)

    def val_step(self, *args, **kwargs):
        return self.module.val_step(*args, **kwargs)

    def zero_grad_buckets(self):
        """Called by OptimizerHook instead of optimizer.zero_grad() (one fill per bucket)."""
        self.reducer.zero_grad()

    def reduce_gradients(self):
        self.reducer.finish()
