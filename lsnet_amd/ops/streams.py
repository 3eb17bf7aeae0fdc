"""THE second stream of a process (one per device).

The training step hands work that is off its critical path to a second stream: the target assignment beside the head's forward
(models/dense_heads/ls_head.py), the rebuild of the stale weight images beside the frozen stem of the next forward (ops/conv.py).
All of it uses ONE stream per device, created as early as possible, and never one stream per module: the HIP runtime multiplexes
streams onto a few hardware queues in creation order, and a stream that lands on the queue of the stream the step runs on
serialises with it -- measured: with one stream per LSHead instance the step of the 2nd, 3rd, 7th model of a process ran 50 - 54 ms
instead of 31 - 32 (profiles/r5_stream_queues.txt), whatever the stream's priority.  The first stream a process creates after the
default one has its own queue.

Rules for whoever puts work on this stream:
  * begin with `side.wait_stream(main)` -- or, where overlap with already queued main-stream work is the point (ops/conv.py: the
    weight-image rebuild behind the optimizer step), with `side.wait_event(ev)` of an event that covers every WRITER of what the
    work reads; that caller checks tensor versions and storage pointers so that nothing but the library's own optimizer kernel
    can have written the weights since the event;
  * tensors allocated under `torch.cuda.stream(side)` belong to the side stream's allocator pool: they are handed to the main
    stream only behind `main.wait_stream(side)`, and their memory returns to the pool when the next side-stream block starts with
    `wait_stream(main)` again (LSHead.forward_train) -- no `record_stream` needed under this rule, and a block that begins with
    `wait_event` must not allocate at all (the rebuild writes into long-lived image buffers);
  * main-stream tensors READ on the side stream while the main stream runs on are protected with `record_stream`
    (ops/resblock.py `_on_side`)."""
import os

import torch

_side = {}
CLAIM_QUEUE = os.environ.get('LSNET_SIDE_STREAM_CLAIM', '1') != '0'   # (0: the A/B arm of tools/rccl_streams.py)


def side_stream(device=None):
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _side.get(idx)
    if st is None:
        st = _side[idx] = torch.cuda.Stream(torch.device('cuda', idx))
        if CLAIM_QUEUE:
            # What decides which streams share a hardware queue is the order of their first SUBMISSIONS, not of their creation
            # (profiles/r6_rccl_streams.txt, default queue count: of ten orders of {this stream, the library's, RCCL's} only those in
            # which this stream submits first and the library's second run the step at 32 ms; DataParallelModel's round-5
            # arrangement -- create this stream, warm the library's, then the first broadcast -- was one of the 49 ms orders because
            # this stream had submitted nothing by then).  One empty-handed kernel now.  (Two hardware queues --
            # GPU_MAX_HW_QUEUES=2 -- make every order fast, but hipGraph replay segfaults with them: lsnet_amd/__init__.py.)
            with torch.cuda.stream(st):
                torch.zeros(1, device=torch.device('cuda', idx)).add_(1.0)
            st.synchronize()
    return st


_warm = set()


def warm_library_streams(device):
    """The library has a second stream of its own (csrc/dcn.hip: the anchor lists of a deformable backward are built beside its
    GEMM), created at the first deformable backward of a process -- in a data-parallel process that is after RCCL has created its
    streams, i.e. at a position in the creation order nobody has measured.  One tiny deformable forward + backward (a 256 -> 256
    3x3 DCNv2 on an 8 x 8 map of zeros: the tower's launch kinds) creates it NOW; DataParallelModel calls this right after
    `side_stream()` and before its first collective, which gives every process the creation order of the single-GPU run.  Never
    raises: a failure here must not keep a model from being built."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _warm:
        return
    _warm.add(idx)
    try:
        from .dcn import modulated_deform_conv
        cl = torch.channels_last
        dev = torch.device('cuda', idx)
        with torch.enable_grad():
            x = torch.zeros(1, 256, 8, 8, device=dev).contiguous(memory_format=cl).requires_grad_()
            w = torch.zeros(256, 256, 3, 3, device=dev).contiguous(memory_format=cl).requires_grad_()
            off = torch.zeros(1, 18, 8, 8, device=dev).contiguous(memory_format=cl).requires_grad_()
            msk = torch.ones(1, 9, 8, 8, device=dev).contiguous(memory_format=cl)
            modulated_deform_conv(x, off, msk, w, None, 1, 1, 1).sum().backward()
    except Exception as ex:   # noqa: BLE001
        import warnings
        warnings.warn(f'lsnet_amd: the warm-up of the library\'s second stream failed ({type(ex).__name__}: {ex}); continuing',
                      RuntimeWarning)
