"""Corner pooling (CornerNet) and the TL / BR pooling blocks of the corner-point-verification head
(mmdet/ops/corner_pool/corner_pool.py:7-177).

`CornerPool(mode)`: running maximum towards one image border -- 'top' pools upwards (out[y] = max over rows >= y),
'bottom' downwards, 'left' / 'right' likewise along x.  The reference's compiled `corner_pool_ext` is dead code on
torch >= 1.5 (it takes the `torch.cummax` branch, :93-102); a scan is a single pass over the map, so this op stays on
the framework's scan primitive as well -- there is no extension module to replace for it."""
import torch
import torch.nn as nn

from ..cnn.bricks import ConvModule
from .conv import Conv2d
from .group_norm import GroupNorm


class CornerPool(nn.Module):

    DIM_FLIP = {'bottom': (2, False), 'left': (3, True), 'right': (3, False), 'top': (2, True)}

    def __init__(self, mode):
        super().__init__()
        assert mode in self.DIM_FLIP
        self.mode = mode

    def forward(self, x):
        dim, flip = self.DIM_FLIP[self.mode]
        if flip:
            x = x.flip(dim)
        out = torch.cummax(x, dim=dim)[0]
        return out.flip(dim) if flip else out


class CornerPoolPack(nn.Module):
    """Two pooled branches (summed, 3x3 conv, GN) + a 1x1 skip branch (conv, GN) -> ReLU -> ConvModule."""

    def __init__(self, dim, pool1, pool2, conv_cfg=None, norm_cfg=None, first_kernel_size=3, kernel_size=3,
                 corner_dim=128):
        super().__init__()
        k1 = first_kernel_size
        self.p1_conv1 = ConvModule(dim, corner_dim, k1, stride=1, padding=(k1 - 1) // 2, conv_cfg=conv_cfg,
                                   norm_cfg=norm_cfg)
        self.p2_conv1 = ConvModule(dim, corner_dim, k1, stride=1, padding=(k1 - 1) // 2, conv_cfg=conv_cfg,
                                   norm_cfg=norm_cfg)
        self.p_conv1 = Conv2d(corner_dim, dim, 3, padding=1, bias=False)
        self.p_gn1 = GroupNorm(num_groups=32, num_channels=dim)
        self.conv1 = Conv2d(dim, dim, 1, bias=False)
        self.gn1 = GroupNorm(num_groups=32, num_channels=dim)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv2 = ConvModule(dim, dim, kernel_size, stride=1, padding=(kernel_size - 1) // 2, conv_cfg=conv_cfg,
                                norm_cfg=norm_cfg)
        self.pool1, self.pool2 = pool1, pool2

    def forward(self, x):
        pooled = self.pool1(self.p1_conv1(x)) + self.pool2(self.p2_conv1(x))
        return self.conv2(self.relu1(self.p_gn1(self.p_conv1(pooled)) + self.gn1(self.conv1(x))))


class TLPool(CornerPoolPack):

    def __init__(self, dim, conv_cfg=None, norm_cfg=None, first_kernel_size=3, kernel_size=3, corner_dim=128):
        super().__init__(dim, CornerPool('top'), CornerPool('left'), conv_cfg, norm_cfg, first_kernel_size,
                         kernel_size, corner_dim)


class BRPool(CornerPoolPack):

    def __init__(self, dim, conv_cfg=None, norm_cfg=None, first_kernel_size=3, kernel_size=3, corner_dim=128):
        super().__init__(dim, CornerPool('bottom'), CornerPool('right'), conv_cfg, norm_cfg, first_kernel_size,
                         kernel_size, corner_dim)
