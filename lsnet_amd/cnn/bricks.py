"""Layer factories and ConvModule -- the subset of mmcv.cnn the LSNet hot path uses
(mmcv/cnn/bricks/{conv,norm,activation,conv_module}.py, mmcv/cnn/utils/weight_init.py)."""
import math
import warnings

import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..ops.conv import Conv2d
from ..ops.group_norm import GroupNorm
from .registry import ACTIVATION_LAYERS, CONV_LAYERS, NORM_LAYERS

CONV_LAYERS.register_module('Conv2d', module=Conv2d)
CONV_LAYERS.register_module('Conv', module=Conv2d)
NORM_LAYERS.register_module('BN', module=nn.BatchNorm2d)
NORM_LAYERS.register_module('BN2d', module=nn.BatchNorm2d)
NORM_LAYERS.register_module('SyncBN', module=nn.SyncBatchNorm)
NORM_LAYERS.register_module('GN', module=GroupNorm)
for _act in (nn.ReLU, nn.LeakyReLU, nn.PReLU, nn.ReLU6, nn.ELU, nn.Sigmoid, nn.Tanh):
    ACTIVATION_LAYERS.register_module(module=_act)

_NORM_ABBR = {'BN': 'bn', 'BN2d': 'bn', 'SyncBN': 'bn', 'GN': 'gn'}


def _split_cfg(cfg, registry, what):
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict')
    if 'type' not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    args = dict(cfg)
    kind = args.pop('type')
    if kind not in registry:
        raise KeyError(f'Unrecognized {what} type {kind}')
    return registry.get(kind), kind, args


def build_conv_layer(cfg, *args, **kwargs):
    """cfg None -> nn.Conv2d; dict(type='DCN'|'DCNv2', ...) -> the deformable packs."""
    cls, _, extra = _split_cfg(dict(type='Conv2d') if cfg is None else cfg, CONV_LAYERS, 'conv')
    return cls(*args, **kwargs, **extra)


def build_norm_layer(cfg, num_features, postfix=''):
    """Returns (name, layer); name = 'bn'/'gn' + postfix.  eps defaults to 1e-5, `requires_grad`
    freezes the affine parameters (mmcv/cnn/bricks/norm.py:70-118)."""
    cls, kind, args = _split_cfg(cfg, NORM_LAYERS, 'norm')
    assert isinstance(postfix, (int, str))
    requires_grad = args.pop('requires_grad', True)
    args.setdefault('eps', 1e-5)
    if kind == 'GN':
        assert 'num_groups' in args
        layer = cls(num_channels=num_features, **args)
    else:
        layer = cls(num_features, **args)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return _NORM_ABBR.get(kind, 'norm') + str(postfix), layer


def build_activation_layer(cfg):
    cls, _, args = _split_cfg(cfg, ACTIVATION_LAYERS, 'activation')
    return cls(**args)


# ---- weight initialisers (mmcv/cnn/utils/weight_init.py:6-66) ------------------------------
def _set_bias(module, bias):
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    _set_bias(module, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    _set_bias(module, bias)


def uniform_init(module, a=0, b=1, bias=0):
    nn.init.uniform_(module.weight, a, b)
    _set_bias(module, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ('uniform', 'normal')
    (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(module.weight, gain=gain)
    _set_bias(module, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    assert distribution in ('uniform', 'normal')
    fn = nn.init.kaiming_uniform_ if distribution == 'uniform' else nn.init.kaiming_normal_
    fn(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    _set_bias(module, bias)


def bias_init_with_prob(prior_prob):
    """bias b with sigmoid(b) == prior_prob"""
    return float(-math.log((1 - prior_prob) / prior_prob))


class ConvModule(nn.Module):
    """conv -> norm -> activation in one block, with the mmcv attribute names (`conv`, `bn`/`gn`,
    `activate`) so checkpoints keep their keys (mmcv/cnn/bricks/conv_module.py:12-200)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 order=('conv', 'norm', 'act')):
        super().__init__()
        assert conv_cfg is None or isinstance(conv_cfg, dict)
        assert norm_cfg is None or isinstance(norm_cfg, dict)
        assert act_cfg is None or isinstance(act_cfg, dict)
        assert isinstance(order, tuple) and set(order) == {'conv', 'norm', 'act'}
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.inplace, self.order = inplace, order
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        self.with_bias = (not self.with_norm) if bias == 'auto' else bias
        if self.with_norm and self.with_bias:
            warnings.warn('ConvModule has norm and bias at the same time')
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=self.with_bias)
        for attr in ('in_channels', 'out_channels', 'kernel_size', 'stride', 'dilation', 'transposed',
                     'output_padding', 'groups'):
            setattr(self, attr, getattr(self.conv, attr))
        self.padding = padding
        if self.with_norm:
            chans = out_channels if order.index('norm') > order.index('conv') else in_channels
            self.norm_name, norm = build_norm_layer(norm_cfg, chans)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act = dict(act_cfg)
            if act['type'] not in ('Tanh', 'PReLU', 'Sigmoid'):
                act.setdefault('inplace', inplace)
            self.activate = build_activation_layer(act)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def init_weights(self):
        if not hasattr(self.conv, 'init_weights'):
            leaky = self.with_activation and self.act_cfg['type'] == 'LeakyReLU'
            kaiming_init(self.conv, a=self.act_cfg.get('negative_slope', 0.01) if leaky else 0,
                         nonlinearity='leaky_relu' if leaky else 'relu')
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def forward(self, x, activate=True, norm=True):
        if (self.order == ('conv', 'norm', 'act') and norm and activate and self.with_norm and self.with_activation
                and isinstance(self.norm, GroupNorm) and isinstance(self.activate, nn.ReLU)):
            return self.norm.forward_act(self.conv(x))      # GroupNorm + ReLU in one pass
        for step in self.order:
            if step == 'conv':
                x = self.conv(x)
            elif step == 'norm' and norm and self.with_norm:
                x = self.norm(x)
            elif step == 'act' and activate and self.with_activation:
                x = self.activate(x)
        return x


__all__ = ['ConvModule', 'build_conv_layer', 'build_norm_layer', 'build_activation_layer', 'constant_init',
           'normal_init', 'uniform_init', 'xavier_init', 'kaiming_init', 'bias_init_with_prob', '_BatchNorm']
