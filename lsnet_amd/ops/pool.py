"""Average pooling with both passes on dense NCHW tensors.

ATen's avg_pool2d BACKWARD for channels-last tensors is wrong on this stack (torch 2.10 + ROCm 7): kernel 3 / stride 2 /
padding 1 returns a gradient off by 0.68 of its range, for a dense channels-last input and for a strided channel slice
alike, while the NCHW kernel agrees with the host bit for bit (tests/test_ops_gpu.py::test_avg_pool_backward).  Found by
the round-4 Res2Net gradient fixture (tools/dbg_res2net.py): the pooled scale of every layerN.0 Bottle2neck was off by 0.86
of its range on the device while the three convolved scales agreed with the host to 8e-7.  The reference's Res2Net runs
its pools in NCHW (mmdet/models/backbones/res2net.py:80-99, 213-231); so do these."""
import torch
import torch.nn.functional as F


def _pair(v):
    return list(v) if isinstance(v, (tuple, list)) else [v, v]


class _AvgPoolNCHW(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, k, s, p, ceil_mode, count_include_pad):
        xc = x.contiguous()
        ctx.cfg = (k, s, p, ceil_mode, count_include_pad)
        ctx.save_for_backward(xc)
        return F.avg_pool2d(xc, k, s, p, ceil_mode, count_include_pad)

    @staticmethod
    def backward(ctx, go):
        (xc,) = ctx.saved_tensors
        k, s, p, ceil_mode, count_include_pad = ctx.cfg
        gx = torch.ops.aten.avg_pool2d_backward(go.contiguous(), xc, _pair(k), _pair(s), _pair(p), ceil_mode,
                                                count_include_pad, None)
        return gx, None, None, None, None, None


def avg_pool_nchw(x, pool):
    """`pool(x)` for an nn.AvgPool2d module, forward and backward through the NCHW kernels."""
    s = pool.stride if pool.stride is not None else pool.kernel_size
    return _AvgPoolNCHW.apply(x, pool.kernel_size, s, pool.padding, bool(pool.ceil_mode), bool(pool.count_include_pad))
