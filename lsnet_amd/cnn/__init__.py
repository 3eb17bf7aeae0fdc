from .bricks import (ConvModule, bias_init_with_prob, build_activation_layer, build_conv_layer, build_norm_layer,
                     constant_init, kaiming_init, normal_init, uniform_init, xavier_init)
from .registry import ACTIVATION_LAYERS, CONV_LAYERS, NORM_LAYERS, PADDING_LAYERS, UPSAMPLE_LAYERS

__all__ = ['ACTIVATION_LAYERS', 'CONV_LAYERS', 'NORM_LAYERS', 'PADDING_LAYERS', 'UPSAMPLE_LAYERS', 'ConvModule',
           'build_conv_layer', 'build_norm_layer', 'build_activation_layer', 'constant_init', 'normal_init',
           'uniform_init', 'xavier_init', 'kaiming_init', 'bias_init_with_prob']
