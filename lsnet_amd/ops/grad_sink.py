"""Parameter-gradient sinks: let the kernel that produces a parameter's gradient ADD it straight into the buffer the
optimizer (and the all-reduce) will read, instead of handing a fresh tensor to autograd.

With `BucketedGradReducer` every `p.grad` is a view into a flat bucket that is zeroed once per step.  Returning a new
`grad_weight` from a custom autograd Function then costs, per parameter and step, an allocation, a memset (the weight-
gradient kernels accumulate over pixel ranges), and autograd's `p.grad.add_(grad_weight)`.  The operators of this package
ask `sink(p)` instead: when the parameter's `.grad` IS its registered sink they call the `accumulate = 1` form of the
C entry point on it (include/lsnet_hip.h: lsn_conv2d_backward_weight ...) and return None for that input; `done(p)`
tells the owner of the sink (the reducer counts a parameter's uses per step to launch bucket all-reduces early).
Anything else -- no reducer, `p.grad` replaced by the user, an operator without sink support -- keeps the classic path,
and the two mix freely because both ADD into the same buffer."""


def register(p, view, on_done=None):
    """`view` is where p's gradient lives for the coming backward passes (the caller zeroes it per step)."""
    p._lsn_sink = view
    p._lsn_sink_done = on_done


def unregister(p):
    for k in ('_lsn_sink', '_lsn_sink_done'):
        if hasattr(p, k):
            delattr(p, k)


def sink(p):
    """The tensor to accumulate p's gradient into, or None."""
    s = getattr(p, '_lsn_sink', None)
    if s is None or p.grad is not s:
        return None
    return s


def done(p):
    cb = getattr(p, '_lsn_sink_done', None)
    if cb is not None:
        cb(p)
