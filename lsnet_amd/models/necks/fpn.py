"""Feature pyramid network (mmdet/models/necks/fpn.py:65-217): 1x1 laterals, nearest top-down
addition, 3x3 output convs, and stride-2 extra levels taken from the last backbone map
('on_input'), the last lateral ('on_lateral') or the last output ('on_output').
Module names (`lateral_convs.i.conv`, `fpn_convs.i.conv`/`.gn`) follow the reference."""
import torch.nn as nn
import torch.nn.functional as F

from ...cnn import ConvModule, xavier_init
from ..builder import NECKS


@NECKS.register_module()
class FPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest')):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        self.in_channels, self.out_channels = list(in_channels), out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.no_norm_on_lateral = no_norm_on_lateral
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:  # no extra level beyond the chosen backbone range
            self.backbone_end_level = end_level
            assert end_level <= self.num_ins
            assert num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        elif add_extra_convs:
            add_extra_convs = 'on_input' if extra_convs_on_inputs else 'on_output'
        self.add_extra_convs = add_extra_convs

        common = dict(conv_cfg=conv_cfg, act_cfg=act_cfg, inplace=False)
        self.lateral_convs = nn.ModuleList(
            ConvModule(self.in_channels[i], out_channels, 1,
                       norm_cfg=None if no_norm_on_lateral else norm_cfg, **common)
            for i in range(start_level, self.backbone_end_level))
        self.fpn_convs = nn.ModuleList(
            ConvModule(out_channels, out_channels, 3, padding=1, norm_cfg=norm_cfg, **common)
            for _ in range(start_level, self.backbone_end_level))
        extra = num_outs - self.backbone_end_level + start_level
        if self.add_extra_convs and extra >= 1:
            for i in range(extra):
                src = self.in_channels[self.backbone_end_level - 1] \
                    if (i == 0 and self.add_extra_convs == 'on_input') else out_channels
                self.fpn_convs.append(ConvModule(src, out_channels, 3, stride=2, padding=1, norm_cfg=norm_cfg,
                                                 **common))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        n = len(lat)
        for i in range(n - 1, 0, -1):  # top-down pathway
            if 'scale_factor' in self.upsample_cfg:
                up = F.interpolate(lat[i], **self.upsample_cfg)
            else:
                up = F.interpolate(lat[i], size=lat[i - 1].shape[2:], **self.upsample_cfg)
            lat[i - 1] = lat[i - 1] + up
        outs = [self.fpn_convs[i](lat[i]) for i in range(n)]
        if self.num_outs > n:
            if not self.add_extra_convs:
                for _ in range(self.num_outs - n):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                src = {'on_input': inputs[self.backbone_end_level - 1], 'on_lateral': lat[-1],
                       'on_output': outs[-1]}[self.add_extra_convs]
                outs.append(self.fpn_convs[n](src))
                for i in range(n + 1, self.num_outs):
                    prev = F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]
                    outs.append(self.fpn_convs[i](prev))
        return tuple(outs)
