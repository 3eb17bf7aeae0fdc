"""THE second stream of a process (one per device).

The training step hands work that is off its critical path to a second stream: the target assignment beside the head's forward
(models/dense_heads/ls_head.py), the rebuild of the stale weight images beside the frozen stem of the next forward (ops/conv.py).
All of it uses ONE stream per device, created as early as possible, and never one stream per module: the HIP runtime multiplexes
streams onto a few hardware queues in creation order, and a stream that lands on the queue of the stream the step runs on
serialises with it -- measured: with one stream per LSHead instance the step of the 2nd, 3rd, 7th model of a process ran 50 - 54 ms
instead of 31 - 32 (profiles/r5_stream_queues.txt), whatever the stream's priority.  The first stream a process creates after the
default one has its own queue."""
import torch

_side = {}


def side_stream(device=None):
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _side.get(idx)
    if st is None:
        st = _side[idx] = torch.cuda.Stream(torch.device('cuda', idx))
    return st
