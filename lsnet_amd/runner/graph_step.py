"""hipGraph capture of the forward + backward pass of one training iteration.

The LSNet step launches ~5 000 kernels, most of them a few microseconds long (per-level head glue, targets,
losses): on MI355X the host cannot feed them fast enough and the GPU idles between launches.  Because the hot
path has no device->host synchronisation and static shapes (fixed image size, dense per-point targets), the
whole of `model.train_step(data)` + `loss.backward()` is captured once into a HIP graph and replayed.

What stays outside the graph, on purpose:
  * the gradient all-reduce (RCCL): gradients are accumulated by the captured backward directly into the flat
    communication buckets (`p.grad` is a view into its bucket), so after the replay the buckets are reduced as
    they are -- no copies, and no collective inside a graph;
  * gradient clipping and the optimizer step (a dozen foreach kernels whose learning rate changes every iteration).
The reference has no counterpart (mmcv's runner launches every kernel eagerly, hooks/optimizer.py:20-28)."""
import torch
import torch.distributed as dist

from ..ops import grad_sink
from ..parallel.reducer import BucketedGradReducer


def _walk(obj, fn, path=()):
    """Applies fn(path, tensor) to every tensor in nested dict / list / tuple containers."""
    if isinstance(obj, torch.Tensor):
        fn(path, obj)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            _walk(v, fn, path + (k,))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            _walk(v, fn, path + (i,))


class GraphedForwardBackward:
    """callable(data) -> outputs of `model.train_step`, with all parameter gradients populated.

    The first `warmup` calls run eagerly (library autotuning, lazy initialisation); the next call captures;
    later calls copy the batch into the captured input tensors (skipped when the caller passes the very same
    tensors) and replay.  A batch whose tensor shapes differ from the captured ones raises: shapes are part of
    the graph."""

    def __init__(self, model, bucket_mb=64.0, warmup=3, process_group=None):
        self.model = model
        self.module = model.module if hasattr(model, 'module') else model
        self.module._log_reduce_elsewhere = True   # __call__ averages the log scalars once per step (base.py _parse_losses)
        self.warmup, self.calls = warmup, 0
        self.graph, self.static_in, self.static_out = None, None, None
        # autograd remembers the stream each parameter's AccumulateGrad node first ran on; the eager warm-up
        # passes therefore run on the very side stream the capture will use
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.params = [p for p in self.module.parameters() if p.requires_grad]
        # flat gradient buckets in reverse registration order (the order backward produces them); p.grad views
        self.buckets = []
        cap = int(bucket_mb * 1024 * 1024 / 4)
        cur, size = [], 0
        for p in reversed(self.params):
            if cur and size + p.numel() > cap:
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self._close(cur)

    def _close(self, plist):
        slot = BucketedGradReducer._slot    # 16-byte aligned gradient sinks (parallel/reducer.py)
        flat = torch.zeros(sum(slot(p.numel()) for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        o = 0
        for p in plist:
            seg = flat[o:o + p.numel()]
            # same memory layout as the parameter (channels-last conv weights): autograd then adds in place
            # without a layout conversion; the all-reduce only sees the flat bucket
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            p.grad = seg.as_strided(p.shape, p.stride()) if dense else seg.view_as(p)
            grad_sink.register(p, p.grad)    # the weight-gradient kernels add straight into the bucket
            o += slot(p.numel())
        self.buckets.append(flat)

    # ------------------------------------------------------------------------------------------
    def _fwd_bwd(self, data):
        for b in self.buckets:
            b.zero_()
        out = self.module.train_step(data, None)
        out['loss'].backward()
        return out

    def _check_grad_views(self):
        o_b = iter(self.buckets)
        for p in self.params:
            assert p.grad is not None and p.grad._base is not None, 'a gradient left its communication bucket'

    def _capture(self, data):
        self.static_in = data
        # every prepared weight image is rebuilt INSIDE the capture (ops/conv.py): a replay then always starts from the
        # current parameters, whether or not an optimizer step ran between the last warm-up call and this one
        from ..ops.conv import invalidate_weight_images
        invalidate_weight_images()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.static_out = self._fwd_bwd(data)
        self._check_grad_views()
        self._sig = []
        _walk(data, lambda path, t: self._sig.append((path, tuple(t.shape), t.dtype)))

    def _bind(self, data):
        if data is self.static_in:
            return
        new = []
        _walk(data, lambda path, t: new.append((path, t)))
        if [(p, tuple(t.shape), t.dtype) for p, t in new] != self._sig:
            raise RuntimeError('GraphedForwardBackward: the batch does not have the captured tensor shapes')
        old = []
        _walk(self.static_in, lambda path, t: old.append(t))
        for dst, (_, src) in zip(old, new):
            if dst is not src:
                dst.copy_(src, non_blocking=True)

    def __call__(self, data):
        self.calls += 1
        if self.graph is None:
            if self.stream is None:
                out = self._fwd_bwd(data)
            elif self.calls <= self.warmup:
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    out = self._fwd_bwd(data)
                torch.cuda.current_stream().wait_stream(self.stream)
            else:
                self._capture(data)
                self.graph.replay()
                out = self.static_out
        else:
            self._bind(data)
            self.graph.replay()
            out = self.static_out
        self.reduce_gradients()
        out = dict(out)
        out['backward_done'] = True
        if self.world > 1 and hasattr(out.get('log_vars'), 'keys'):
            # no collective is captured (base.py skips it while capturing): the log scalars are averaged here
            lv = out['log_vars']
            keys = list(lv.keys())
            packed = torch.stack([lv[k].detach().reshape(()) for k in keys]).div_(self.world)
            dist.all_reduce(packed, group=self.group)
            avg = type(lv)()
            for i, k in enumerate(keys):
                avg[k] = packed[i]
            out['log_vars'] = avg
        return out

    def reduce_gradients(self):
        """Average the buckets over the data-parallel ranks (eager RCCL all-reduce, one per bucket)."""
        if self.world == 1:
            return
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets]
        for w, b in zip(works, self.buckets):
            w.wait()
            b.div_(self.world)
