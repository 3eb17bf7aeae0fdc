"""Res2Net backbone (mmdet/models/backbones/res2net.py:17-350), SURVEY.md section 8f rank 3: the backbone of the
reference's headline configs (`lsnet_*_res2_101_fpn_dconv_c3-c5_*`).

A Bottle2neck splits the 1x1-expanded features into `scales` groups of `width` channels; group i is convolved (3x3,
DCNv2 in the dconv stages) after adding the previous group's output, the last group passes through (or through a 3x3
average pool in the first block of a stage), everything is concatenated and projected back.  Deep stem (three 3x3
convs) and average-pool shortcuts are always on, as in the reference (`Res2Net.__init__` overrides both).

Parameter names match the reference (`convs.i`, `bns.i`, `conv1/bn1`, `conv3/bn3`, `stem.*`), so its checkpoints load."""
import math

import torch
import torch.nn as nn
import torch.utils.checkpoint as cp

from ...cnn import build_conv_layer, build_norm_layer, constant_init
from ...ops.batch_norm import bn_act
from ...ops.pool import avg_pool_nchw
from ..builder import BACKBONES
from .resnet import Bottleneck, ResNet, _shortcut


class Bottle2neck(Bottleneck):
    expansion = 4

    def __init__(self, inplanes, planes, scales=4, base_width=26, base_channels=64, stage_type='normal', **kwargs):
        super().__init__(inplanes, planes, **kwargs)
        assert scales > 1, 'Res2Net degenerates to ResNet when scales = 1.'
        width = int(math.floor(self.planes * (base_width / base_channels)))
        self.norm1_name, norm1 = build_norm_layer(self.norm_cfg, width * scales, postfix=1)
        self.norm3_name, norm3 = build_norm_layer(self.norm_cfg, self.planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(self.conv_cfg, self.inplanes, width * scales, kernel_size=1,
                                      stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        if stage_type == 'stage' and self.conv2_stride != 1:
            self.pool = nn.AvgPool2d(kernel_size=3, stride=self.conv2_stride, padding=1)
        conv_cfg = self.conv_cfg
        if self.with_dcn:
            dcn = dict(self.dcn)
            if not dcn.pop('fallback_on_stride', False):
                assert self.conv_cfg is None, 'conv_cfg must be None for DCN'
                conv_cfg = dcn
        self.convs = nn.ModuleList(
            build_conv_layer(conv_cfg, width, width, kernel_size=3, stride=self.conv2_stride, padding=self.dilation,
                             dilation=self.dilation, bias=False) for _ in range(scales - 1))
        self.bns = nn.ModuleList(build_norm_layer(self.norm_cfg, width, postfix=i + 1)[1] for i in range(scales - 1))
        self.conv3 = build_conv_layer(self.conv_cfg, width * scales, self.planes * self.expansion, kernel_size=1,
                                      bias=False)
        self.add_module(self.norm3_name, norm3)
        self.stage_type, self.scales, self.width = stage_type, scales, width
        del self.conv2
        delattr(self, self.norm2_name)

    def _body(self, x):
        out = bn_act(self.norm1, self.conv1(x), relu=True)
        spx = torch.split(out, self.width, 1)
        parts, sp = [], None
        for i in range(self.scales - 1):
            sp = spx[i] if (i == 0 or self.stage_type == 'stage') else sp + spx[i]
            sp = bn_act(self.bns[i], self.convs[i](sp.contiguous(memory_format=_fmt(sp))), relu=True)
            parts.append(sp)
        last = spx[self.scales - 1]
        if not (self.stage_type == 'normal' or self.conv2_stride == 1):
            last = avg_pool_nchw(last, self.pool)    # (ops/pool.py: ATen's channels-last backward is wrong on this stack)
        out = torch.cat(parts + [last], 1)
        return bn_act(self.norm3, self.conv3(out), relu=True, residual=_shortcut(self.downsample, x))

    def forward(self, x):
        return cp.checkpoint(self._body, x) if (self.with_cp and x.requires_grad) else self._body(x)


def _fmt(t):
    """keep the memory format the network runs in (a channel split of a channels-last tensor is a strided view)"""
    return torch.channels_last if t.stride(1) == 1 and t.dim() == 4 and t.shape[1] > 1 else torch.contiguous_format


class Res2Layer(nn.Sequential):
    """One stage: first block of type 'stage' with the average-pool + 1x1 shortcut, the rest 'normal'."""

    def __init__(self, block, inplanes, planes, num_blocks, stride=1, avg_down=True, conv_cfg=None,
                 norm_cfg=dict(type='BN'), scales=4, base_width=26, **kwargs):
        self.block = block
        downsample = None
        out_planes = planes * block.expansion
        if stride != 1 or inplanes != out_planes:
            downsample = nn.Sequential(
                nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True, count_include_pad=False),
                build_conv_layer(conv_cfg, inplanes, out_planes, kernel_size=1, stride=1, bias=False),
                build_norm_layer(norm_cfg, out_planes)[1])
        blocks = [block(inplanes=inplanes, planes=planes, stride=stride, downsample=downsample, conv_cfg=conv_cfg,
                        norm_cfg=norm_cfg, scales=scales, base_width=base_width, stage_type='stage', **kwargs)]
        blocks += [block(inplanes=out_planes, planes=planes, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                         scales=scales, base_width=base_width, **kwargs) for _ in range(1, num_blocks)]
        super().__init__(*blocks)


@BACKBONES.register_module()
class Res2Net(ResNet):
    arch_settings = {50: (Bottle2neck, (3, 4, 6, 3)), 101: (Bottle2neck, (3, 4, 23, 3)),
                     152: (Bottle2neck, (3, 8, 36, 3))}

    def __init__(self, scales=4, base_width=26, style='pytorch', deep_stem=True, avg_down=True, **kwargs):
        self.scales, self.base_width = scales, base_width
        super().__init__(style='pytorch', deep_stem=True, avg_down=True, **kwargs)

    def make_res_layer(self, **kwargs):
        kwargs.pop('plugins', None)
        return Res2Layer(scales=self.scales, base_width=self.base_width, base_channels=self.base_channels, **kwargs)

    def init_weights(self, pretrained=None):
        super().init_weights(pretrained)
        if isinstance(pretrained, str):
            return
        if self.dcn is not None:
            for m in self.modules():
                if isinstance(m, Bottle2neck):
                    for c in m.convs:
                        if hasattr(c, 'conv_offset'):
                            constant_init(c.conv_offset, 0)
