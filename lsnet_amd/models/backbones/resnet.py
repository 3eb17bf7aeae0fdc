"""ResNet / ResNeXt backbones with optional DCN in conv2 (mmdet/models/backbones/resnet.py:261-301,
303-661; resnext.py:11-131; utils/res_layer.py:24-102).

Module and parameter names are the reference's (`conv1`, `bn1`, `layerN.M.conv2.conv_offset`,
`downsample.0/1`) so torchvision / reference checkpoints load unchanged.  One Bottleneck class
serves both families: ResNeXt is the grouped-width case."""
import math

import torch
import torch.nn as nn
import torch.utils.checkpoint as cp
from torch.nn.modules.batchnorm import _BatchNorm

from ...cnn import build_conv_layer, build_norm_layer, constant_init, kaiming_init
from ...ops.batch_norm import bn_act
from ...ops import resblock
from ...ops.conv import conv_bn_act, conv_bn_act_frozen
from ...ops.pool import avg_pool_nchw
from ..builder import BACKBONES


def _conv_bn(conv, bn, x, relu, residual=None):
    """act(bn(conv(x)) + residual): one fused forward launch for a dense convolution + eval-mode BatchNorm pair -- frozen
    (ops/conv.py conv_bn_act_frozen: no backward at all) or trainable (conv_bn_act: the norm folded into the weight image
    of the step) -- else the convolution followed by the fused norm + add + ReLU pass (deformable conv2, GroupNorm,
    training-mode statistics)."""
    out = None
    if isinstance(bn, _BatchNorm):
        out = conv_bn_act_frozen(conv, bn, x, relu, residual)
        if out is None:
            out = conv_bn_act(conv, bn, x, relu, residual)
    return out if out is not None else bn_act(bn, conv(x), relu=relu, residual=residual)


def _shortcut(downsample, x):
    """identity, or the projection shortcut conv -> norm (its norm through the fused kernel as well)."""
    if downsample is None:
        return x
    mods = list(downsample)
    if isinstance(mods[-1], _BatchNorm):
        for m in mods[:-2]:
            x = avg_pool_nchw(x, m) if isinstance(m, nn.AvgPool2d) else m(x)   # (avg_down shortcuts: ops/pool.py)
        if len(mods) >= 2:
            return _conv_bn(mods[-2], mods[-1], x, relu=False)
        return bn_act(mods[-1], x, relu=False)
    return downsample(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, plugins=None, **_):
        super().__init__()
        assert dcn is None and plugins is None, 'Not implemented yet.'
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=dilation,
                                      dilation=dilation, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride, self.dilation, self.with_cp = downsample, stride, dilation, with_cp

    norm1 = property(lambda self: getattr(self, self.norm1_name))
    norm2 = property(lambda self: getattr(self, self.norm2_name))

    def _body(self, x):
        out = _conv_bn(self.conv1, self.norm1, x, relu=True)
        return _conv_bn(self.conv2, self.norm2, out, relu=True, residual=_shortcut(self.downsample, x))

    def forward(self, x):
        return cp.checkpoint(self._body, x) if (self.with_cp and x.requires_grad) else self._body(x)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (stride here for style='pytorch'; DCN here when `dcn` is given) -> 1x1, + identity.
    groups > 1 gives the ResNeXt block with width floor(planes * base_width / base_channels) * groups."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, plugins=None, groups=1, base_width=4,
                 base_channels=64):
        super().__init__()
        assert style in ('pytorch', 'caffe')
        assert dcn is None or isinstance(dcn, dict)
        if plugins:
            raise NotImplementedError('backbone plugins are outside the LSNet hot path')
        self.inplanes, self.planes, self.stride, self.dilation = inplanes, planes, stride, dilation
        self.style, self.with_cp, self.conv_cfg, self.norm_cfg = style, with_cp, conv_cfg, norm_cfg
        self.dcn, self.with_dcn = dcn, dcn is not None
        self.conv1_stride, self.conv2_stride = (1, stride) if style == 'pytorch' else (stride, 1)
        width = planes if groups == 1 else math.floor(planes * (base_width / base_channels)) * groups

        self.norm1_name, norm1 = build_norm_layer(norm_cfg, width, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, width, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, width, kernel_size=1, stride=self.conv1_stride,
                                      bias=False)
        self.add_module(self.norm1_name, norm1)
        conv2_cfg = conv_cfg
        if self.with_dcn:
            dcn = dict(dcn)
            if not dcn.pop('fallback_on_stride', False):
                assert conv_cfg is None, 'conv_cfg must be None for DCN'
                conv2_cfg = dcn
        self.conv2 = build_conv_layer(conv2_cfg, width, width, kernel_size=3, stride=self.conv2_stride,
                                      padding=dilation, dilation=dilation, groups=groups, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, width, planes * self.expansion, kernel_size=1, bias=False)
        self.add_module(self.norm3_name, norm3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    norm1 = property(lambda self: getattr(self, self.norm1_name))
    norm2 = property(lambda self: getattr(self, self.norm2_name))
    norm3 = property(lambda self: getattr(self, self.norm3_name))

    def _body(self, x):
        out = _conv_bn(self.conv1, self.norm1, x, relu=True)
        out = _conv_bn(self.conv2, self.norm2, out, relu=True)
        # relu(bn3(conv3) + identity): norm, residual add and activation in one pass (resnet.py:261-301)
        return _conv_bn(self.conv3, self.norm3, out, relu=True, residual=_shortcut(self.downsample, x))

    def forward(self, x, pregate_in=False, gy_pregated=False, wg_queue=None, wg_flush=False, fused=None):
        """pregate_in / gy_pregated: set by ResLayer.forward for neighbouring blocks that both run as the fused autograd
        node of ops/resblock.py (the ReLU gate of a block's output then rides in the NEXT block's backward-data launch).
        wg_queue / wg_flush: the stage's identical blocks share their weight-gradient launches (ops/resblock.py).
        fused: ResLayer.forward's decision for THIS block's input (it pairs the flags above by it); None: decide here."""
        if fused is None:
            fused = fused_block_ok(self, x)
        if fused:
            return resblock.bottleneck(self, x.contiguous(memory_format=torch.channels_last), pregate_in, gy_pregated,
                                       wg_queue, wg_flush)
        assert not (pregate_in or gy_pregated or wg_queue is not None), 'ResLayer.forward pairs the flags of fused blocks only'
        return cp.checkpoint(self._body, x) if (self.with_cp and x.requires_grad) else self._body(x)


def fused_block_ok(blk, x, numel=None):
    """The whole bottleneck as one autograd node (ops/resblock.py): a trainable dense block behind eval-mode norms, on a
    device fp32 tensor small enough for 32-bit byte offsets in every map of the block (the widest is 4 x the input).
    numel: the element count of the block's OWN input when x is only a tensor of the same kind (ResLayer.forward decides
    for every block of the stage before any of them has run)."""
    n = x.numel() if numel is None and torch.is_tensor(x) else numel
    return (type(blk) is Bottleneck and torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
            and n * 16 < 2 ** 31 and resblock.bottleneck_ok(blk))


class ResLayer(nn.Sequential):
    """One stage: the first block carries the stride and the projection shortcut."""

    def forward(self, x):
        blocks = list(self)
        # Every block is judged on ITS OWN input (blocks 1 .. n-1 of a stage see block 0's output -- in a trainable
        # stage 1 four times the stage input -- and a block that declined after its neighbours had been told otherwise hit an
        # assertion): block 0 maps (B, C, H, W) to (B, 4 planes, ceil(H / s), ceil(W / s)), the others keep that shape.
        numel = [x.numel()] * len(blocks)
        if torch.is_tensor(x) and x.dim() == 4 and len(blocks) > 1:
            b0 = blocks[0]
            s = getattr(b0, 'stride', 1)
            s = s if isinstance(s, int) else s[0]
            co = getattr(getattr(b0, 'conv3', None), 'out_channels', None) or getattr(getattr(b0, 'conv2', None), 'out_channels', x.shape[1])
            n_out = x.shape[0] * co * (-(-x.shape[2] // s)) * (-(-x.shape[3] // s))
            numel = [x.numel()] + [n_out] * (len(blocks) - 1)
        ok = [fused_block_ok(b, x, n) for b, n in zip(blocks, numel)]
        # blocks 1 .. n-1 are built alike (__init__ below), and block 0's conv3 has their conv3's geometry: when the whole
        # stage runs fused, the weight gradients of one geometry are ONE launch, issued by block 0's backward -- the last
        group = len(blocks) > 2 and all(ok) and torch.is_grad_enabled()
        queue = resblock.WgQueue() if group else None
        for i, b in enumerate(blocks):
            if ok[i]:
                x = b(x, pregate_in=i > 0 and ok[i - 1], gy_pregated=i + 1 < len(blocks) and ok[i + 1],
                      wg_queue=queue, wg_flush=group and i == 0, fused=True)
            else:
                x = b(x, fused=False) if type(b) is Bottleneck else b(x)
        return x

    def __init__(self, block, inplanes, planes, num_blocks, stride=1, avg_down=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), **kwargs):
        self.block = block
        downsample = None
        out_planes = planes * block.expansion
        if stride != 1 or inplanes != out_planes:
            mods, conv_stride = [], stride
            if avg_down and stride != 1:
                conv_stride = 1
                mods.append(nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True,
                                         count_include_pad=False))
            mods += [build_conv_layer(conv_cfg, inplanes, out_planes, kernel_size=1, stride=conv_stride,
                                      bias=False),
                     build_norm_layer(norm_cfg, out_planes)[1]]
            downsample = nn.Sequential(*mods)
        blocks = [block(inplanes=inplanes, planes=planes, stride=stride, downsample=downsample, conv_cfg=conv_cfg,
                        norm_cfg=norm_cfg, **kwargs)]
        blocks += [block(inplanes=out_planes, planes=planes, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                         **kwargs) for _ in range(1, num_blocks)]
        super().__init__(*blocks)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {
        18: (BasicBlock, (2, 2, 2, 2)),
        34: (BasicBlock, (3, 4, 6, 3)),
        50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)),
        152: (Bottleneck, (3, 8, 36, 3)),
    }

    def __init__(self, depth, in_channels=3, stem_channels=64, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        assert 1 <= num_stages <= 4
        assert len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages
        if dcn is not None:
            assert len(stage_with_dcn) == num_stages
        self.depth, self.stem_channels, self.base_channels = depth, stem_channels, base_channels
        self.num_stages, self.strides, self.dilations = num_stages, strides, dilations
        self.out_indices, self.style, self.avg_down, self.deep_stem = out_indices, style, avg_down, deep_stem
        self.frozen_stages, self.conv_cfg, self.norm_cfg = frozen_stages, conv_cfg, norm_cfg
        self.with_cp, self.norm_eval, self.dcn, self.stage_with_dcn = with_cp, norm_eval, dcn, stage_with_dcn
        self.zero_init_residual = zero_init_residual
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]

        if deep_stem:   # ResNetV1d / Res2Net stem: three 3x3 convs (resnet.py:521-556), module name `stem`
            half = stem_channels // 2
            mods = []
            for cin, cout, st in ((in_channels, half, 2), (half, half, 1), (half, stem_channels, 1)):
                mods += [build_conv_layer(conv_cfg, cin, cout, kernel_size=3, stride=st, padding=1, bias=False),
                         build_norm_layer(norm_cfg, cout)[1], nn.ReLU(inplace=True)]
            self.stem = nn.Sequential(*mods)
        else:
            self.conv1 = build_conv_layer(conv_cfg, in_channels, stem_channels, kernel_size=7, stride=2, padding=3,
                                          bias=False)
            self.norm1_name, norm1 = build_norm_layer(norm_cfg, stem_channels, postfix=1)
            self.add_module(self.norm1_name, norm1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        self.res_layers, inplanes = [], stem_channels
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = base_channels * 2 ** i
            layer = self.make_res_layer(block=self.block, inplanes=inplanes, planes=planes,
                                        num_blocks=num_blocks, stride=strides[i], dilation=dilations[i],
                                        style=style, avg_down=avg_down, with_cp=with_cp, conv_cfg=conv_cfg,
                                        norm_cfg=norm_cfg, dcn=dcn if stage_with_dcn[i] else None, plugins=None)
            inplanes = planes * self.block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, layer)
            self.res_layers.append(name)
        self._freeze_stages()
        self.feat_dim = self.block.expansion * base_channels * 2 ** (len(self.stage_blocks) - 1)

    def make_res_layer(self, **kwargs):
        return ResLayer(**kwargs)

    norm1 = property(lambda self: getattr(self, self.norm1_name))

    def _freeze_stages(self):
        """Stem (frozen_stages >= 0) and the first `frozen_stages` stages: eval mode, no grads."""
        frozen = []
        if self.frozen_stages >= 0:
            if self.deep_stem:
                self.stem.eval()
                frozen.append(self.stem)
            else:
                self.norm1.eval()
                frozen += [self.conv1, self.norm1]
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            frozen.append(m)
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from ...runner.checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
            return
        if pretrained is not None:
            raise TypeError('pretrained must be a str or None')
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                kaiming_init(m)
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                constant_init(m, 1)
        if self.dcn is not None:
            for m in self.modules():
                if isinstance(m, Bottleneck) and hasattr(getattr(m, 'conv2', None), 'conv_offset'):   # (Bottle2neck: `convs`)
                    constant_init(m.conv2.conv_offset, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    constant_init(m.norm3, 0)
                elif isinstance(m, BasicBlock):
                    constant_init(m.norm2, 0)

    def forward(self, x):
        x = self.stem(x) if self.deep_stem else _conv_bn(self.conv1, self.norm1, x, relu=True)
        x = self.maxpool(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self


@BACKBONES.register_module()
class ResNeXt(ResNet):
    arch_settings = {
        50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)),
        152: (Bottleneck, (3, 8, 36, 3)),
    }

    def __init__(self, groups=1, base_width=4, **kwargs):
        self.groups, self.base_width = groups, base_width
        super().__init__(**kwargs)

    def make_res_layer(self, **kwargs):
        return ResLayer(groups=self.groups, base_width=self.base_width, base_channels=self.base_channels,
                        **kwargs)
